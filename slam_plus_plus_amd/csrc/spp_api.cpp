// spp_api.cpp -- the extern "C" boundary of libspp_hip.so (include/spp_hip.h) and the ctx plumbing.
// No compute happens here; every numeric entry point ends in a HIP kernel launch on ctx->stream
// and fails loudly (SPP_E_NO_DEVICE / SPP_E_HIP) when there is no usable gfx950 device.

#include "spp_internal.h"
#include <string.h>
#include <new>

using namespace spp;

namespace spp {

void phases_reset(spp_ctx *ctx)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE))
		return;
	if(!ctx->timer.created) {
		for(int i = 0; i < 2 * SPP_N_PHASES; ++ i)
			SPP_HIP_CHECK(hipEventCreate(&ctx->timer.ev[i]));
		ctx->timer.created = true;
	}
	for(int i = 0; i < SPP_N_PHASES; ++ i) {
		ctx->timer.used[i] = false;
		ctx->phase_ms[i] = 0;
	}
	ctx->dom_used = 0;
	ctx->dom_flops = 0;
}

void phase_begin(spp_ctx *ctx, int phase)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE) || !ctx->timer.created)
		return;
	SPP_HIP_CHECK(hipEventRecord(ctx->timer.ev[2 * phase], ctx->stream));
}

void phase_end(spp_ctx *ctx, int phase)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE) || !ctx->timer.created)
		return;
	SPP_HIP_CHECK(hipEventRecord(ctx->timer.ev[2 * phase + 1], ctx->stream));
	ctx->timer.used[phase] = true;
}

void phases_collect(spp_ctx *ctx)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE) || !ctx->timer.created)
		return;
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	for(int i = 0; i < SPP_N_PHASES; ++ i) {
		if(!ctx->timer.used[i])
			continue;
		float ms = 0;
		if(hipEventElapsedTime(&ms, ctx->timer.ev[2 * i], ctx->timer.ev[2 * i + 1]) == hipSuccess)
			ctx->phase_ms[i] = ms;
	}
}

void dom_begin(spp_ctx *ctx)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE))
		return;
	if(ctx->dom_used + 2 > ctx->dom_events.size()) {
		size_t old = ctx->dom_events.size();
		ctx->dom_events.resize(old + 64);
		for(size_t i = old; i < ctx->dom_events.size(); ++ i)
			SPP_HIP_CHECK(hipEventCreate(&ctx->dom_events[i]));
	}
	SPP_HIP_CHECK(hipEventRecord(ctx->dom_events[ctx->dom_used], ctx->stream));
}

void dom_end(spp_ctx *ctx, double flops)
{
	if(!(ctx->flags & SPP_FLAG_PROFILE))
		return;
	SPP_HIP_CHECK(hipEventRecord(ctx->dom_events[ctx->dom_used + 1], ctx->stream));
	ctx->dom_used += 2;
	ctx->dom_flops += flops;
}

} // namespace spp

#define SPP_TRY(ctx) try {
#define SPP_CATCH(ctx) } catch(const spp::Error &e) { if(ctx) (ctx)->last_error = e.what(); return e.code; } \
	catch(const std::bad_alloc &) { if(ctx) (ctx)->last_error = "host allocation failed"; return SPP_E_NOMEM; } \
	catch(const std::exception &e) { if(ctx) (ctx)->last_error = e.what(); return SPP_E_HIP; }

extern "C" {

const char *spp_version(void)
{
	return "slam_plus_plus_amd 0.1 (gfx950, fp64)";
}

spp_ctx *spp_create(int device, int flags)
{
	int count = 0;
	if(hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
		return nullptr; // no CPU fallback: the adapter turns this into std::runtime_error
	if(hipSetDevice(device) != hipSuccess)
		return nullptr;
	spp_ctx *ctx = new(std::nothrow) spp_ctx;
	if(!ctx)
		return nullptr;
	ctx->device = device;
	ctx->flags = flags;
	if(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
		delete ctx;
		return nullptr;
	}
	ctx->own_stream = true;
	return ctx;
}

int spp_free_memory(spp_ctx *ctx)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	ctx->schur.release_all();
	sparse_release(ctx);
	assemble_release(ctx);
	ctx->dense.tinv.release();
	ctx->dense.tinv_all.release();
	ctx->dense.xtmp.release();
	ctx->dense.la_cnt.release();
	ctx->dense.la_trace.release();
	ctx->geom_partial.release();
	ctx->dense.flags.release();
	ctx->dense.trsv_pay.release();
	ctx->dense.trsv_m.release();
	ctx->dense.tail_pub.release();
	ctx->dense.tail_order.release();
	ctx->dense.tail_order_tr = ctx->dense.tail_order_tc = 0;
	ctx->dense.tail_dinv.release();
	ctx->dense.tail_epoch = 0;
	ctx->dense.epoch = 0;
	ctx->dense.sync.release();
	ctx->dense.fuse_cnt.release();
	ctx->dense.fuse_expect.clear();
	ctx->dense.sync_epoch = 0;
	ctx->d_vals.release();
	ctx->d_rhs.release();
	ctx->mode = -1;
	return SPP_OK;
	SPP_CATCH(ctx)
}

void spp_destroy(spp_ctx *ctx)
{
	if(!ctx)
		return;
	if(ctx->plan_trash.joinable())
		ctx->plan_trash.join();
	spp_free_memory(ctx);
	ctx->dense.info.release();
	if(ctx->dense.aux) {
		dense_aux_park(ctx->device, ctx->dense.aux); // (kept for the next context of this device: creating it is ~10 ms)
		(void)hipEventDestroy(ctx->dense.ev[0]);
		(void)hipEventDestroy(ctx->dense.ev[1]);
	}
	if(ctx->dense.chain) {
		(void)hipStreamDestroy(ctx->dense.chain);
		if(ctx->dense.ev_chain)
			(void)hipEventDestroy(ctx->dense.ev_chain);
	}
	if(ctx->dense.h_chain_err)
		(void)hipHostFree(ctx->dense.h_chain_err);
	if(ctx->h_staging)
		(void)hipHostFree(ctx->h_staging);
	if(ctx->timer.created)
		for(int i = 0; i < 2 * SPP_N_PHASES; ++ i)
			(void)hipEventDestroy(ctx->timer.ev[i]);
	for(size_t i = 0; i < ctx->dom_events.size(); ++ i)
		(void)hipEventDestroy(ctx->dom_events[i]);
	if(ctx->own_stream && ctx->stream)
		(void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int spp_last_error(const spp_ctx *ctx, char *buf, size_t buf_size)
{
	if(!ctx || !buf || !buf_size)
		return SPP_E_BADARG;
	strncpy(buf, ctx->last_error.c_str(), buf_size - 1);
	buf[buf_size - 1] = 0;
	return SPP_OK;
}

double *spp_host_staging(spp_ctx *ctx, int64_t n_doubles)
{
	if(!ctx || n_doubles < 0)
		return nullptr;
	if((size_t)n_doubles <= ctx->h_staging_cap && ctx->h_staging)
		return ctx->h_staging;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream); // a copy out of the old buffer may still be in flight
	if(ctx->h_staging) {
		(void)hipHostFree(ctx->h_staging);
		ctx->h_staging = nullptr;
		ctx->h_staging_cap = 0;
	}
	const size_t want = (size_t)(n_doubles > 0 ? n_doubles : 1);
	if(hipHostMalloc((void**)&ctx->h_staging, want * sizeof(double), hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		ctx->h_staging = nullptr;
		ctx->last_error = "hipHostMalloc of the staging buffer failed";
		return nullptr;
	}
	ctx->h_staging_cap = want;
	return ctx->h_staging;
}

int spp_set_stream(spp_ctx *ctx, void *hip_stream)
{
	if(!ctx)
		return SPP_E_BADARG;
	(void)hipStreamSynchronize(ctx->stream);
	if(ctx->own_stream && ctx->stream)
		(void)hipStreamDestroy(ctx->stream);
	ctx->stream = (hipStream_t)hip_stream;
	ctx->own_stream = false;
	return SPP_OK;
}

int spp_synchronize(spp_ctx *ctx)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	dense_chain_check(ctx);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_set_profiling(spp_ctx *ctx, int on)
{
	if(!ctx)
		return SPP_E_BADARG;
	ctx->flags = on ? (ctx->flags | SPP_FLAG_PROFILE) : (ctx->flags & ~SPP_FLAG_PROFILE);
	return SPP_OK;
}

int spp_set_shard(spp_ctx *ctx, int rank, int world_size)
{
	if(!ctx || world_size < 1 || rank < 0 || rank >= world_size)
		return SPP_E_BADARG;
	ctx->shard_rank = rank;
	ctx->shard_world = world_size;
	return SPP_OK;
}

int spp_analyze(spp_ctx *ctx, int64_t nb, const int64_t *col_ptr, const int64_t *row_idx,
	const int64_t *blk_off, const int32_t *dim, int mode)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(nb > 0 && col_ptr && row_idx && blk_off && dim, SPP_E_BADARG, "spp_analyze: null or empty structure");
	SPP_REQUIRE(col_ptr[0] == 0 && col_ptr[nb] >= nb, SPP_E_BADARG, "spp_analyze: column pointers must start at 0 and hold every diagonal block");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	// The structure is validated into a local object and only then committed: a rejected call must not leave the
	// sizes of the NEW structure beside the plan of the OLD one (the next solve would copy the wrong extents).
	// The old plan is dropped first in any case -- after a failed analyze the ctx is "not analyzed".
	ctx->mode = -1;
	VClock clk("spp_analyze");
	Structure st;
	st.nb = nb;
	st.nnzb = col_ptr[nb];
	st.col_ptr.assign(col_ptr, col_ptr + nb + 1);
	SPP_REQUIRE(st.nnzb >= nb, SPP_E_BADARG, "spp_analyze: fewer blocks than block columns");
	for(int64_t j = 0; j < nb; ++ j)
		SPP_REQUIRE(col_ptr[j + 1] > col_ptr[j], SPP_E_BADARG, "every block column needs its diagonal block");
	st.dim.assign(dim, dim + nb);
	st.base.resize(nb + 1);
	st.base[0] = 0;
	for(int64_t j = 0; j < nb; ++ j) {
		// 7 = Sim(3) poses (include/slam/Sim3_Types.h); the sparse path works on scalar fronts and only needs a block to
		// fit one wave in the assembly step (8 x 8 = 64 lanes); the Schur kernels are instantiated for {6,3}, {3,2}, 3, 6
		SPP_REQUIRE(dim[j] > 0 && dim[j] <= 8, SPP_E_BADARG, "block widths must be in 1..8");
		st.base[j + 1] = st.base[j] + dim[j];
	}
	st.n = st.base[nb];
	// the block lists are copied and checked by ranges of columns on host threads (54 MB on a Venice-sized structure)
	st.row_idx.resize(st.nnzb);
	st.blk_off.resize(st.nnzb);
	{
		const int nt = plan_threads(st.nnzb);
		std::vector<int64_t> jcut, nvals_t(nt, 0);
		balanced_cuts(st.col_ptr, nt, jcut);
		run_threads(nt, [&](int t) {
			const int64_t p0 = col_ptr[jcut[t]], p1 = col_ptr[jcut[t + 1]];
			if(p1 > p0) {
				memcpy(st.row_idx.data() + p0, row_idx + p0, (size_t)(p1 - p0) * sizeof(int64_t));
				memcpy(st.blk_off.data() + p0, blk_off + p0, (size_t)(p1 - p0) * sizeof(int64_t));
			}
			int64_t nv = 0;
			for(int64_t j = jcut[t]; j < jcut[t + 1]; ++ j) {
				for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
					const int64_t i = row_idx[p];
					SPP_REQUIRE(i >= 0 && i <= j, SPP_E_BADARG, "only the upper triangle may be stored");
					SPP_REQUIRE(p == col_ptr[j] || row_idx[p - 1] < i, SPP_E_BADARG, "rows must ascend within a column");
					SPP_REQUIRE(blk_off[p] >= 0, SPP_E_BADARG, "negative block offset");
					nv = std::max<int64_t>(nv, blk_off[p] + (int64_t)dim[i] * dim[j]);
				}
				SPP_REQUIRE(row_idx[col_ptr[j + 1] - 1] == j, SPP_E_BADARG, "diagonal block missing (must be last in its column)");
			}
			nvals_t[t] = nv;
		});
		st.nvals = 0;
		for(int t = 0; t < nt; ++ t)
			st.nvals = std::max(st.nvals, nvals_t[t]);
	}
	ctx->st = std::move(st);
	Structure &st_ref = ctx->st;
	clk.lap("structure copied + validated");
	ctx->schur.release_all();
	sparse_release(ctx);
	clk.lap("old plans released");
	int dp, dl;
	int chosen = mode;
	if(mode == SPP_MODE_AUTO) {
		chosen = SPP_MODE_SPARSE;
		if(schur_applicable(st_ref, &dp, &dl)) {
			// dense reduced system while it is small enough for the MFMA dense factor to win (Venice: 5226),
			// sparse (supernodal) reduced system beyond: a 10k-camera S is 28.8 GB dense and mostly zeros
			int64_t n_red = 0;
			for(int64_t j = 0; j < nb; ++ j)
				n_red += (dim[j] == dp) ? dp : 0;
			chosen = (n_red <= 16384) ? SPP_MODE_SCHUR : SPP_MODE_SCHUR_SPARSE;
		}
	}
	SPP_REQUIRE(chosen == SPP_MODE_SCHUR || chosen == SPP_MODE_SPARSE || chosen == SPP_MODE_SCHUR_SPARSE ||
		chosen == SPP_MODE_SCHUR_MIS, SPP_E_BADARG, "unknown mode");
	SPP_REQUIRE((chosen != SPP_MODE_SPARSE && chosen != SPP_MODE_SCHUR_MIS) || ctx->shard_world == 1, SPP_E_UNSUPPORTED,
		"only the Schur modes shard (over landmarks); pose graphs run as replicas (DESIGN.md, multi-GPU)");
	ctx->mode = -1;
	if(chosen == SPP_MODE_SCHUR || chosen == SPP_MODE_SCHUR_SPARSE || chosen == SPP_MODE_SCHUR_MIS) {
		const bool mis = chosen == SPP_MODE_SCHUR_MIS;
		const bool sparse_S = chosen != SPP_MODE_SCHUR; // the reduced system of a pose graph is sparse
		build_schur_plan(ctx, sparse_S, mis);
		clk.lap("schur plan (all of it)");
		ctx->schur.mis = mis;
		ctx->order.clear();
		ctx->order.reserve((size_t)st_ref.nb);
		if(sparse_S) {
			sparse_analyze(ctx, ctx->schur.s_st); // leaves the elimination order of the reduced poses in ctx->order
			std::vector<int64_t> red = ctx->order;
			ctx->order.clear();
			for(size_t i = 0; i < red.size(); ++ i)
				ctx->order.push_back(ctx->schur.pose_block[red[i]]);
			chosen = SPP_MODE_SCHUR; // one internal mode; SPP_INFO_MODE reports the variant
		} else
		for(size_t i = 0; i < ctx->schur.pose_block.size(); ++ i)
			ctx->order.push_back(ctx->schur.pose_block[i]);
		// landmarks are eliminated FIRST in elimination terms; the reference lists them last in its
		// guided ordering (poses | landmarks), which is what order[] reports
		for(int64_t j = 0; j < st_ref.nb; ++ j)
			if(ctx->schur.is_lm[j])
				ctx->order.push_back(j);
	} else
		sparse_analyze(ctx, ctx->st);
	ctx->mode = chosen;
	clk.lap("plan + ordering");
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_get_info(const spp_ctx *ctx, int what, int64_t *out)
{
	if(!ctx || !out)
		return SPP_E_BADARG;
	if(ctx->mode < 0 && what != SPP_INFO_NNZB && what != SPP_INFO_NVALS && what != SPP_INFO_N)
		return SPP_E_STATE;
	switch(what) {
	case SPP_INFO_MODE: *out = (ctx->mode != SPP_MODE_SCHUR) ? ctx->mode : ctx->schur.mis ? SPP_MODE_SCHUR_MIS :
		ctx->schur.sparse_S ? SPP_MODE_SCHUR_SPARSE : SPP_MODE_SCHUR; break;
	case SPP_INFO_N: *out = ctx->st.n; break;
	case SPP_INFO_NNZB: *out = ctx->st.nnzb; break;
	case SPP_INFO_NVALS: *out = ctx->st.nvals; break;
	case SPP_INFO_FACTOR_NNZ: *out = ctx->factor_nnz; break;
	case SPP_INFO_FACTOR_FLOPS: *out = ctx->factor_flops; break;
	case SPP_INFO_N_REDUCED: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.n_red : 0; break;
	case SPP_INFO_N_POSES: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.nc : ctx->st.nb; break;
	case SPP_INFO_N_LANDMARKS: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.nl : 0; break;
	case SPP_INFO_SCHUR_PAIRS: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.n_pairs : 0; break;
	case SPP_INFO_N_OBS: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.no : 0; break;
	case SPP_INFO_SOLVE_BYTES: *out = ctx->solve_bytes; break;
	case SPP_INFO_N_SUPERNODES: *out = ctx->sparse ? sparse_info(ctx, what) : 0; break;
	case SPP_INFO_N_LEVELS: *out = ctx->sparse ? sparse_info(ctx, what) : 0; break;
	case SPP_INFO_S_LD: *out = (ctx->mode == SPP_MODE_SCHUR && !ctx->schur.sparse_S) ? ctx->schur.ld : 0; break;
	case SPP_INFO_S_NNZB: *out = (ctx->mode == SPP_MODE_SCHUR) ? ctx->schur.n_sblk : 0; break;
	case SPP_INFO_DENSE_STREAMED: *out = ctx->dense.tail_rows_last; break;
	default: return SPP_E_BADARG;
	}
	return SPP_OK;
}

int spp_get_ordering(const spp_ctx *ctx, int64_t *h_order)
{
	if(!ctx || !h_order)
		return SPP_E_BADARG;
	if(ctx->mode < 0)
		return SPP_E_STATE;
	for(size_t i = 0; i < ctx->order.size(); ++ i)
		h_order[i] = ctx->order[i];
	return SPP_OK;
}

int spp_block_ordering(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, int method, int64_t *h_order)
{
	if(nb <= 0 || !col_ptr || !row_idx || !h_order || (method != SPP_ORDER_AMD && method != SPP_ORDER_ND))
		return SPP_E_BADARG;
	try {
		for(int64_t j = 0; j < nb; ++ j)
			for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p)
				if(row_idx[p] < 0 || row_idx[p] > j)
					return SPP_E_BADARG; // upper triangle only
		std::vector<int64_t> order;
		if(method == SPP_ORDER_AMD)
			min_degree_order(nb, col_ptr, row_idx, order);
		else
			nested_dissection_order(nb, col_ptr, row_idx, order);
		for(int64_t k = 0; k < nb; ++ k)
			h_order[k] = order[k];
		return SPP_OK;
	} catch(const std::bad_alloc &) {
		return SPP_E_NOMEM;
	} catch(...) {
		return SPP_E_HIP;
	}
}

int spp_schur_plan_host(int64_t nb, const int32_t *dim, const int64_t *col_ptr, const int64_t *row_idx, int shard_rank,
	int shard_world, int sparse_S, int64_t *out, double *seconds)
{
	if(nb <= 0 || !dim || !col_ptr || !row_idx || !out || shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world)
		return SPP_E_BADARG;
	try {
		Structure st;
		st.nb = nb;
		st.nnzb = col_ptr[nb];
		st.col_ptr.assign(col_ptr, col_ptr + nb + 1);
		st.row_idx.assign(row_idx, row_idx + st.nnzb);
		st.dim.assign(dim, dim + nb);
		st.base.resize(nb + 1);
		st.base[0] = 0;
		for(int64_t j = 0; j < nb; ++ j)
			st.base[j + 1] = st.base[j] + dim[j];
		st.n = st.base[nb];
		st.blk_off.resize(st.nnzb);
		int64_t off = 0;
		for(int64_t j = 0; j < nb; ++ j)
			for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
				if(row_idx[p] < 0 || row_idx[p] > j)
					return SPP_E_BADARG; // upper triangle only
				st.blk_off[p] = off;
				off += (int64_t)dim[row_idx[p]] * dim[j];
			}
		st.nvals = off;
		const double sec = schur_plan_host_probe(st, shard_rank, shard_world, sparse_S != 0, out);
		if(seconds)
			*seconds = sec;
		return SPP_OK;
	} catch(const Error &e) {
		return e.code;
	} catch(const std::bad_alloc &) {
		return SPP_E_NOMEM;
	} catch(...) {
		return SPP_E_HIP;
	}
}

int spp_schur_buffer_size(const spp_ctx *ctx, int64_t *n_doubles)
{
	if(!ctx || !n_doubles)
		return SPP_E_BADARG;
	if(ctx->mode != SPP_MODE_SCHUR)
		return SPP_E_STATE;
	*n_doubles = schur_buffer_doubles(ctx);
	return SPP_OK;
}

int spp_schur_form(spp_ctx *ctx, const double *d_vals, const double *d_rhs, double *d_S_rhs)
{
	if(!ctx || !d_vals || !d_rhs || !d_S_rhs)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode == SPP_MODE_SCHUR, SPP_E_STATE, "spp_schur_form: analyze in Schur mode first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	phases_reset(ctx);
	phase_begin(ctx, SPP_PHASE_TOTAL);
	schur_form(ctx, d_vals, d_rhs, d_S_rhs);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_schur_packed_size(const spp_ctx *ctx, int64_t *n_doubles)
{
	if(!ctx || !n_doubles)
		return SPP_E_BADARG;
	if(ctx->mode != SPP_MODE_SCHUR)
		return SPP_E_STATE;
	if(ctx->schur.sparse_S) { // already compact: block values | rhs
		*n_doubles = schur_buffer_doubles(ctx);
		return SPP_OK;
	}
	const int64_t nblk = ctx->schur.ld / DENSE_NB;
	*n_doubles = (int64_t)DENSE_NB * DENSE_NB * (nblk * (nblk + 1) / 2);
	return SPP_OK;
}

int spp_schur_pack(spp_ctx *ctx, const double *d_S_rhs, double *d_packed)
{
	if(!ctx || !d_S_rhs || !d_packed)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode == SPP_MODE_SCHUR, SPP_E_STATE, "spp_schur_pack: analyze in Schur mode first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	if(ctx->schur.sparse_S)
		SPP_HIP_CHECK(hipMemcpyAsync(d_packed, d_S_rhs, (size_t)schur_buffer_doubles(ctx) * sizeof(double),
			hipMemcpyDeviceToDevice, ctx->stream));
	else
		schur_pack(ctx, const_cast<double*>(d_S_rhs), d_packed, true);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_schur_unpack(spp_ctx *ctx, const double *d_packed, double *d_S_rhs)
{
	if(!ctx || !d_S_rhs || !d_packed)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode == SPP_MODE_SCHUR, SPP_E_STATE, "spp_schur_unpack: analyze in Schur mode first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	if(ctx->schur.sparse_S)
		SPP_HIP_CHECK(hipMemcpyAsync(d_S_rhs, d_packed, (size_t)schur_buffer_doubles(ctx) * sizeof(double),
			hipMemcpyDeviceToDevice, ctx->stream));
	else
		schur_pack(ctx, d_S_rhs, const_cast<double*>(d_packed), false);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_schur_finish(spp_ctx *ctx, const double *d_vals, double *d_S_rhs, double *d_rhs_inout)
{
	if(!ctx || !d_vals || !d_rhs_inout || !d_S_rhs)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode == SPP_MODE_SCHUR, SPP_E_STATE, "spp_schur_finish: analyze in Schur mode first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	int ret = schur_finish(ctx, d_vals, d_S_rhs, d_rhs_inout); // synchronizes (status fetch)
	phase_end(ctx, SPP_PHASE_TOTAL);
	phases_collect(ctx);
	dense_chain_check(ctx); // a timed-out backward substitution is reported by the call that produced the result
	return ret;
	SPP_CATCH(ctx)
}

int spp_factor_solve_device(spp_ctx *ctx, const double *d_vals, double *d_rhs)
{
	if(!ctx || !d_vals || !d_rhs)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode >= 0, SPP_E_STATE, "spp_factor_solve: call spp_analyze first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	phases_reset(ctx);
	phase_begin(ctx, SPP_PHASE_TOTAL);
	int ret;
	if(ctx->mode == SPP_MODE_SCHUR) {
		SPP_REQUIRE(ctx->shard_world == 1, SPP_E_STATE,
			"sharded ctx: use spp_schur_form / all-reduce / spp_schur_finish");
		ctx->schur.S.reserve((size_t)schur_buffer_doubles(ctx));
		schur_form(ctx, d_vals, d_rhs, ctx->schur.S.p);
		ret = schur_finish(ctx, d_vals, ctx->schur.S.p, d_rhs);
	} else
		ret = sparse_factor_solve(ctx, d_vals, d_rhs);
	phase_end(ctx, SPP_PHASE_TOTAL);
	phases_collect(ctx);
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	dense_chain_check(ctx);
	return ret;
	SPP_CATCH(ctx)
}

int spp_factor_solve(spp_ctx *ctx, const double *h_vals, double *h_rhs)
{
	if(!ctx || !h_vals || !h_rhs)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->mode >= 0, SPP_E_STATE, "spp_factor_solve: call spp_analyze first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	ctx->d_vals.reserve((size_t)ctx->st.nvals);
	ctx->d_rhs.reserve((size_t)ctx->st.n);
	SPP_HIP_CHECK(hipMemcpyAsync(ctx->d_vals.p, h_vals, ctx->st.nvals * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	SPP_HIP_CHECK(hipMemcpyAsync(ctx->d_rhs.p, h_rhs, ctx->st.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	int ret = spp_factor_solve_device(ctx, ctx->d_vals.p, ctx->d_rhs.p);
	if(ret != SPP_OK)
		return ret; // the rhs is left untouched on failure, like the reference (LinearSolver_UberBlock.h:411-423)
	SPP_HIP_CHECK(hipMemcpyAsync(h_rhs, ctx->d_rhs.p, ctx->st.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_assemble_analyze(spp_ctx *ctx, int64_t nv, const int32_t *h_dim, int64_t ne, const int64_t *h_v0,
	const int64_t *h_v1, int d0, int d1, int rd, int64_t unary_vertex)
{
	if(!ctx || !h_dim || !h_v0 || !h_v1 || nv <= 0 || ne <= 0)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	assemble_analyze(ctx, nv, h_dim, ne, h_v0, h_v1, d0, d1, rd, unary_vertex);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_assemble_get_structure(const spp_ctx *ctx, int64_t *h_col_ptr, int64_t *h_row_idx, int64_t *h_blk_off)
{
	if(!ctx || !ctx->assemble)
		return SPP_E_STATE;
	assemble_get_structure(ctx, h_col_ptr, h_row_idx, h_blk_off);
	return SPP_OK;
}

int spp_assemble_device(spp_ctx *ctx, const double *d_J0, const double *d_J1, const double *d_Omega,
	const double *d_r, double damping, double *d_vals_out, double *d_eta_out)
{
	if(!ctx || !d_J0 || !d_J1 || !d_Omega || !d_r || !d_vals_out || !d_eta_out)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->assemble, SPP_E_STATE, "spp_assemble_device: call spp_assemble_analyze first");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	phases_reset(ctx);
	phase_begin(ctx, SPP_PHASE_ASSEMBLE);
	assemble_run(ctx, d_J0, d_J1, d_Omega, d_r, damping, d_vals_out, d_eta_out);
	phase_end(ctx, SPP_PHASE_ASSEMBLE);
	phases_collect(ctx);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_assemble_set_edge_weights(spp_ctx *ctx, const double *d_w)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE(ctx->assemble, SPP_E_STATE, "spp_assemble_set_edge_weights: call spp_assemble_analyze first");
	assemble_set_edge_weights(ctx, d_w);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_se2_linearize_device(spp_ctx *ctx, int64_t n_edges, const int32_t *d_v0, const int32_t *d_v1,
	const double *d_poses, const double *d_measurements, double *d_J0, double *d_J1, double *d_r)
{
	if(!ctx || n_edges < 0 || !d_v0 || !d_v1 || !d_poses || !d_measurements || !d_J0 || !d_J1 || !d_r)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	se2_linearize(ctx, n_edges, d_v0, d_v1, d_poses, d_measurements, d_J0, d_J1, d_r);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_se2_update_device(spp_ctx *ctx, int64_t n_vertices, double *d_poses, const double *d_dx, int apply,
	double *h_dx_norm2)
{
	if(!ctx || n_vertices < 0 || !d_poses || !d_dx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	const double n2 = se2_update(ctx, n_vertices, d_poses, d_dx, apply != 0);
	if(h_dx_norm2)
		*h_dx_norm2 = n2;
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_se3_linearize_device(spp_ctx *ctx, int64_t n_edges, const int32_t *d_v0, const int32_t *d_v1,
	const double *d_poses, const double *d_measurements, double *d_J0, double *d_J1, double *d_r)
{
	if(!ctx || n_edges < 0 || !d_v0 || !d_v1 || !d_poses || !d_measurements || !d_J0 || !d_J1 || !d_r)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	se3_linearize(ctx, n_edges, d_v0, d_v1, d_poses, d_measurements, d_J0, d_J1, d_r);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_se3_update_device(spp_ctx *ctx, int64_t n_vertices, double *d_poses, const double *d_dx, int apply,
	double *h_dx_norm2)
{
	if(!ctx || n_vertices < 0 || !d_poses || !d_dx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	const double n2 = se3_update(ctx, n_vertices, d_poses, d_dx, apply != 0);
	if(h_dx_norm2)
		*h_dx_norm2 = n2;
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_ba_linearize_device(spp_ctx *ctx, int64_t n_obs, const int32_t *d_cam_of, const int32_t *d_pt_of,
	const double *d_cams, const double *d_intrinsics, const double *d_points, const double *d_measurements,
	double *d_J0, double *d_J1, double *d_r)
{
	if(!ctx || n_obs < 0 || !d_cam_of || !d_pt_of || !d_cams || !d_intrinsics || !d_points || !d_measurements ||
	   !d_J0 || !d_J1 || !d_r)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	ba_linearize(ctx, n_obs, d_cam_of, d_pt_of, d_cams, d_intrinsics, d_points, d_measurements, d_J0, d_J1, d_r);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_ba_update_device(spp_ctx *ctx, int64_t n_cams, double *d_cams, const int64_t *d_cam_dxoff,
	int64_t n_points, double *d_points, const int64_t *d_pt_dxoff, const double *d_dx, int64_t n_dx, int apply,
	double *h_dx_norm2)
{
	if(!ctx || n_cams < 0 || n_points < 0 || n_dx < 0 || !d_dx || (n_cams && (!d_cams || !d_cam_dxoff)) ||
	   (n_points && (!d_points || !d_pt_dxoff)))
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	const double n2 = ba_update(ctx, n_cams, d_cams, d_cam_dxoff, n_points, d_points, d_pt_dxoff, d_dx, n_dx, apply != 0);
	if(h_dx_norm2)
		*h_dx_norm2 = n2;
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_edge_robust_weights_device(spp_ctx *ctx, int64_t n_edges, int rd, int kind, double scale, double param,
	const double *d_r, double *d_w_out)
{
	if(!ctx || n_edges < 0 || !d_r || !d_w_out)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	edge_robust_weights(ctx, n_edges, rd, kind, scale, param, d_r, d_w_out);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_edge_chi2_device(spp_ctx *ctx, int64_t n_edges, int rd, const double *d_r, const double *d_Omega, double *h_chi2)
{
	if(!ctx || n_edges < 0 || !d_r || !d_Omega || !h_chi2)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*h_chi2 = edge_chi2(ctx, n_edges, rd, d_r, d_Omega);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_edge_hessian_maxdiag_device(spp_ctx *ctx, int64_t n_edges, int rd, int d0, int d1, const double *d_J0,
	const double *d_J1, const double *d_Omega, double *h_max)
{
	if(!ctx || n_edges < 0 || !d_J0 || !d_J1 || !d_Omega || !h_max)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*h_max = edge_hessian_maxdiag(ctx, n_edges, rd, d0, d1, d_J0, d_J1, d_Omega);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_lm_gain_denominator_device(spp_ctx *ctx, int64_t n, const double *d_dx, const double *d_eta, double alpha,
	double *h_out)
{
	if(!ctx || n < 0 || !d_dx || !d_eta || !h_out)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*h_out = lm_gain_denominator(ctx, n, d_dx, d_eta, alpha);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_device_malloc(spp_ctx *ctx, size_t bytes, void **d_ptr)
{
	if(!ctx || !d_ptr)
		return SPP_E_BADARG;
	(void)hipSetDevice(ctx->device);
	if(hipMalloc(d_ptr, bytes ? bytes : 8) != hipSuccess) {
		ctx->last_error = "hipMalloc failed";
		return SPP_E_NOMEM;
	}
	return SPP_OK;
}

int spp_device_free(spp_ctx *ctx, void *d_ptr)
{
	if(!ctx)
		return SPP_E_BADARG;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	return hipFree(d_ptr) == hipSuccess ? SPP_OK : SPP_E_HIP;
}

int spp_memcpy_h2d(spp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	SPP_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_memcpy_d2h(spp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	SPP_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_memcpy_d2d(spp_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
	if(!ctx)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	SPP_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_get_phase_ms(spp_ctx *ctx, double *ms_out)
{
	if(!ctx || !ms_out)
		return SPP_E_BADARG;
	for(int i = 0; i < SPP_N_PHASES; ++ i)
		ms_out[i] = ctx->phase_ms[i];
	return SPP_OK;
}

int spp_get_dominant_kernel(spp_ctx *ctx, double *ms_total, int64_t *n_launches, double *flops)
{
	if(!ctx || !ms_total || !n_launches || !flops)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	double total = 0;
	for(size_t i = 0; i + 1 < ctx->dom_used; i += 2) {
		float ms = 0;
		SPP_HIP_CHECK(hipEventElapsedTime(&ms, ctx->dom_events[i], ctx->dom_events[i + 1]));
		total += ms;
	}
	*ms_total = total;
	*n_launches = (int64_t)(ctx->dom_used / 2);
	*flops = ctx->dom_flops;
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_microbench_copy(spp_ctx *ctx, size_t bytes, int iters, double *gb_per_s)
{
	if(!ctx || !gb_per_s)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*gb_per_s = microbench_copy(ctx, bytes, iters);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_microbench_ctile(spp_ctx *ctx, int n, int iters, double *gb_per_s)
{
	if(!ctx || !gb_per_s)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*gb_per_s = microbench_ctile(ctx, n, iters);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_microbench_update(spp_ctx *ctx, int64_t m, int iters, double *ms_per_launch)
{
	if(!ctx || !ms_per_launch || m < 128 || iters < 1)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*ms_per_launch = microbench_update(ctx, m, iters);
	return SPP_OK;
	SPP_CATCH(ctx)
}

int spp_microbench_mfma_f64(spp_ctx *ctx, int iters, double *tflops)
{
	if(!ctx || !tflops)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	*tflops = microbench_mfma_f64(ctx, iters);
	return SPP_OK;
	SPP_CATCH(ctx)
}

// ---- dense kernels exposed for unit tests: arbitrary n, copies into a padded workspace ----------
int spp_dense_potrf_upper(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld)
{
	if(!ctx || !d_A || n <= 0 || ld < n)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	const int64_t ldp = ((n + 1 + DENSE_NB - 1) / DENSE_NB) * DENSE_NB;
	DevBuf<double> tmp;
	tmp.reserve((size_t)ldp * ldp);
	SPP_HIP_CHECK(hipMemsetAsync(tmp.p, 0, (size_t)ldp * ldp * sizeof(double), ctx->stream));
	SPP_HIP_CHECK(hipMemcpy2DAsync(tmp.p, ldp * sizeof(double), d_A, ld * sizeof(double), n * sizeof(double), n,
		hipMemcpyDeviceToDevice, ctx->stream));
	dense_set_padding(ctx, tmp.p, ldp, n);
	int ret = dense_potrf_upper(ctx, tmp.p, n, ldp, true);
	SPP_HIP_CHECK(hipMemcpy2DAsync(d_A, ld * sizeof(double), tmp.p, ldp * sizeof(double), n * sizeof(double), n,
		hipMemcpyDeviceToDevice, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return ret;
	SPP_CATCH(ctx)
}

int spp_dense_posv(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld, double *d_b)
{
	if(!ctx || !d_A || !d_b || n <= 0 || ld < n)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	const int64_t ldp = ((n + 1 + DENSE_NB - 1) / DENSE_NB) * DENSE_NB;
	DevBuf<double> tmp;
	tmp.reserve((size_t)ldp * ldp);
	SPP_HIP_CHECK(hipMemsetAsync(tmp.p, 0, (size_t)ldp * ldp * sizeof(double), ctx->stream));
	SPP_HIP_CHECK(hipMemcpy2DAsync(tmp.p, ldp * sizeof(double), d_A, ld * sizeof(double), n * sizeof(double), n,
		hipMemcpyDeviceToDevice, ctx->stream));
	dense_set_padding(ctx, tmp.p, ldp, n);
	SPP_HIP_CHECK(hipMemcpyAsync(tmp.p + n * ldp, d_b, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
	int ret = dense_potrf_upper(ctx, tmp.p, n, ldp, true);
	if(ret == SPP_OK) {
		dense_potrs_upper(ctx, tmp.p, n, ldp, tmp.p + n * ldp);
		SPP_HIP_CHECK(hipMemcpyAsync(d_b, tmp.p + n * ldp, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
		SPP_HIP_CHECK(hipMemcpy2DAsync(d_A, ld * sizeof(double), tmp.p, ldp * sizeof(double), n * sizeof(double), n,
			hipMemcpyDeviceToDevice, ctx->stream));
	}
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	dense_chain_check(ctx);
	return ret;
	SPP_CATCH(ctx)
}

int spp_dense_gemm_tn_sub(spp_ctx *ctx, int64_t m, int64_t n, int64_t k, const double *d_A, int64_t lda,
	const double *d_B, int64_t ldb, double *d_C, int64_t ldc)
{
	if(!ctx || !d_A || !d_B || !d_C)
		return SPP_E_BADARG;
	SPP_TRY(ctx)
	SPP_REQUIRE((lda % 2) == 0 && (ldb % 2) == 0, SPP_E_BADARG, "gemm_tn_sub: lda/ldb must be even (16-byte loads)");
	SPP_HIP_CHECK(hipSetDevice(ctx->device));
	dense_gemm_tn_sub(ctx, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, false);
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return SPP_OK;
	SPP_CATCH(ctx)
}

} // extern "C"
