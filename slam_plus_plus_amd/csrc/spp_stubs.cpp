// temporary stubs (replaced by spp_sparse.hip / spp_assemble.hip)
#include "spp_internal.h"
namespace spp {
#ifndef SPP_HAVE_SPARSE
void sparse_analyze(spp_ctx *, const Structure &) { throw Error(SPP_E_UNSUPPORTED, "sparse mode not built"); }
int sparse_factor_solve(spp_ctx *, const double *, double *) { throw Error(SPP_E_UNSUPPORTED, "sparse mode not built"); }
void sparse_release(spp_ctx *) {}
int64_t sparse_info(const spp_ctx *, int) { return 0; }
#endif
#ifndef SPP_HAVE_ASSEMBLE
void assemble_analyze(spp_ctx *, int64_t, const int32_t *, int64_t, const int64_t *, const int64_t *, int, int, int, int64_t) { throw Error(SPP_E_UNSUPPORTED, "assembly not built"); }
void assemble_run(spp_ctx *, const double *, const double *, const double *, const double *, double, double *, double *) { throw Error(SPP_E_UNSUPPORTED, "assembly not built"); }
void assemble_release(spp_ctx *) {}
void assemble_get_structure(const spp_ctx *, int64_t *, int64_t *, int64_t *) {}
#endif
}
