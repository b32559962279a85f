import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else "ba_banded"
prob = synth.make(name)
ctx = api.Context(0, api.FLAG_PROFILE)
st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
d_in = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
d_vals = api.DeviceArray(ctx, st.nvals); d_eta = api.DeviceArray(ctx, st.n); d_rhs = api.DeviceArray(ctx, st.n)
ctx.assemble_device(d_in[0].ptr, d_in[1].ptr, d_in[2].ptr, d_in[3].ptr, prob.damping, d_vals.ptr, d_eta.ptr)
ctx.synchronize()
for mode in (api.MODE_SCHUR_SPARSE, api.MODE_SCHUR):
    if mode == api.MODE_SCHUR and ctx.info("N_REDUCED") > 20000:
        continue
    t0 = time.time(); ctx.analyze(st, mode); ta = time.time() - t0
    print("mode", ctx.info("MODE"), "analyze %.2fs" % ta, "n_red", ctx.info("N_REDUCED"), "S blocks", ctx.info("S_NNZB"),
          "factor nnz", ctx.info("FACTOR_NNZ"), "flops %.3g" % ctx.info("FACTOR_FLOPS"), "supernodes", ctx.info("N_SUPERNODES"), "levels", ctx.info("N_LEVELS"), flush=True)
    for it in range(4):
        d_rhs.copy_from(d_eta)
        t0 = time.perf_counter(); code = ctx.factor_solve_device(d_vals.ptr, d_rhs.ptr); dt = time.perf_counter() - t0
    x = d_rhs.download()
    print("  code", code, "%.3f ms" % (dt * 1e3), {k: round(v, 3) for k, v in ctx.phase_ms().items()}, "|x|", np.linalg.norm(x), flush=True)
    if mode == api.MODE_SCHUR_SPARSE:
        xs = x
    else:
        print("  rel diff sparse-S vs dense-S", np.linalg.norm(xs - x) / np.linalg.norm(x))
