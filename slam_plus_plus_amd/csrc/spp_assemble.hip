// spp_assemble.hip -- Lambda = J^T Omega J and eta = J^T Omega r on gfx950, fp64.
//
// Replaces
//   symbolic: CLambdaOps2::AddEntriesInSparseSystem + Alloc_HessianBlocks_v2
//             (reference include/slam/NonlinearSolver_Lambda_Base.h:1852-1931, include/slam/BaseTypes_Binary.h:525-660)
//   numeric : Refresh_Lambda = Calculate_Hessians_v2 over all edges + CMatrixReductionPlan::ReduceAll
//             + CVectorReductionPlan::ReduceAll (_Lambda_Base.h:1658-1688, BaseTypes_Binary.h:759-848,
//             _Lambda_Base.h:563-607,743-756,152-197,395-399), the unary factor (:1903-1924) and the
//             Levenberg-Marquardt damping (NonlinearSolver_Lambda_LM.h:228-239).
//
// The reference writes every edge's H00 / H11 / g0 / g1 into temporaries and then gather-sums them
// per destination (1.02 GB of temporaries per refresh on Venice). Here the reduction
// plan is kept (same destinations, same edge order) but nothing is materialized: each destination
// block recomputes its contributions straight from J / Omega / r, which are read three times
// (off-diagonal block, vertex 0, vertex 1) and Lambda is written once.
//   offdiag_kernel   one thread per off-diagonal block: sum over its edges of J0^T Omega J1 (or the
//                    transposed form when the vertex ids are reversed, BaseTypes_Binary.h:783-806)
//   vertex_seq_kernel one thread per low-degree vertex, contributions summed sequentially in edge order
//                    (bit-identical to the reference's order: first assigned, rest added)
//   vertex_wave_kernel one wave per high-degree vertex (cameras): lanes stride over the edge list,
//                    fixed butterfly reduction => reproducible, order differs from sequential in the
//                    last bits only
// All three are HBM-bound: 192 B read + 144 B written per BA observation for the off-diagonal part.

#include "spp_internal.h"
#include <algorithm>

namespace spp {

struct AssemblePlan {
	int d0 = 0, d1 = 0, rd = 0;
	int64_t nv = 0, ne = 0, unary_vertex = -1;
	Structure st;
	int64_t n_ob = 0;
	std::vector<int64_t> h_vlist_seq[2], h_vlist_wave[2]; // per vertex-dimension class (d0 / d1)
	DevBuf<unsigned char> index_store; // the one allocation behind the index arrays below (UploadArena)
	DevBuf<int32_t> ob_ptr;     // [n_ob+1]
	DevBuf<int32_t> ob_edge;    // edge | reversed << 31
	DevBuf<int64_t> ob_off;     // [n_ob] offset of the block in vals
	DevBuf<int32_t> vl_ptr;     // [nv+1]
	DevBuf<int32_t> vl_entry;   // edge << 1 | side
	DevBuf<int64_t> v_doff;     // [nv] offset of the diagonal block
	DevBuf<int64_t> v_base;     // [nv] scalar offset in eta
	DevBuf<int32_t> vlist_seq[2], vlist_wave[2];
	int64_t n_seq[2] = {0, 0}, n_wave[2] = {0, 0};
	const double *edge_weights = nullptr; // device, one robust weight per edge, or null (assemble_set_edge_weights; not owned)
};

void assemble_release(spp_ctx *ctx)
{
	delete ctx->assemble;
	ctx->assemble = nullptr;
}

void assemble_get_structure(const spp_ctx *ctx, int64_t *col_ptr, int64_t *row_idx, int64_t *blk_off)
{
	const Structure &st = ctx->assemble->st;
	std::copy(st.col_ptr.begin(), st.col_ptr.end(), col_ptr);
	std::copy(st.row_idx.begin(), st.row_idx.end(), row_idx);
	std::copy(st.blk_off.begin(), st.blk_off.end(), blk_off);
}

void assemble_set_edge_weights(spp_ctx *ctx, const double *d_w)
{
	ctx->assemble->edge_weights = d_w;
}

static const int SEQ_MAX_DEGREE = 24;

void assemble_analyze(spp_ctx *ctx, int64_t nv, const int32_t *dim, int64_t ne, const int64_t *v0,
	const int64_t *v1, int d0, int d1, int rd, int64_t unary_vertex)
{
	SPP_REQUIRE((d0 == 6 && d1 == 3 && rd == 2) || (d0 == 3 && d1 == 3 && rd == 3) ||
		(d0 == 6 && d1 == 6 && rd == 6) || (d0 == 3 && d1 == 2 && rd == 2), SPP_E_UNSUPPORTED,
		"edge group (d0, d1, rd) not instantiated: (6,3,2) (3,3,3) (6,6,6) (3,2,2)");
	SPP_REQUIRE(ne < (int64_t(1) << 30), SPP_E_UNSUPPORTED, "too many edges for 31-bit edge indices");
	assemble_release(ctx); // after a rejected call the ctx has NO assembly plan (spp_assemble_device then fails its state check)
	VClock clk("assemble_analyze");
	const int nt = plan_threads(ne);
	run_threads(nt, [&](int t) {
		for(int64_t e = ne * t / nt, e1 = ne * (t + 1) / nt; e < e1; ++ e) {
			SPP_REQUIRE(v0[e] >= 0 && v0[e] < nv && v1[e] >= 0 && v1[e] < nv && v0[e] != v1[e], SPP_E_BADARG, "bad edge");
			SPP_REQUIRE(dim[v0[e]] == d0 && dim[v1[e]] == d1, SPP_E_BADARG, "vertex width does not match the edge group");
		}
	});
	SPP_REQUIRE(unary_vertex < nv, SPP_E_BADARG, "unary_vertex out of range");
	// built in a local object, installed in the ctx only when complete
	struct PlanGuard { AssemblePlan *p; ~PlanGuard() { delete p; } } guard = {new AssemblePlan};
	AssemblePlan *ap = guard.p;
	ap->d0 = d0; ap->d1 = d1; ap->rd = rd; ap->nv = nv; ap->ne = ne; ap->unary_vertex = unary_vertex;
	// ---- Lambda structure: diagonal of every vertex + upper block of every edge
	// (_Lambda_Base.h:1863-1881 builds all block rows/cols first, then :1897 allocates edge blocks)
	HVec<std::pair<int64_t, int64_t> > key(ne); // (col, row)
	run_threads(nt, [&](int t) {
		for(int64_t e = ne * t / nt, e1 = ne * (t + 1) / nt; e < e1; ++ e)
			key[e] = std::make_pair(std::max(v0[e], v1[e]), std::min(v0[e], v1[e]));
	});
	// edges in (column, row, edge index) order: a counting sort by column, then (row, edge) inside each column -- a handful
	// per landmark column -- by insertion, longer runs by std::sort (a comparison sort of all the edges was most of this
	// function on a Venice-sized graph)
	HVec<int64_t> eorder(ne);
	std::vector<int64_t> cstart(nv + 1, 0); // edges of column c: eorder[cstart[c] .. cstart[c + 1])
	std::vector<int64_t> ccut; // ranges of columns with about equal numbers of edges (+ 1 per column)
	{
		// counted and scattered by all threads with atomic increments; the scatter is then in no particular order, the sort
		// inside each column is by (row, edge index) and restores it
		run_threads(nt, [&](int t) {
			for(int64_t e = ne * t / nt, e1 = ne * (t + 1) / nt; e < e1; ++ e)
				__atomic_fetch_add(&cstart[key[e].first + 1], (int64_t)1, __ATOMIC_RELAXED);
		});
		for(int64_t c = 0; c < nv; ++ c)
			cstart[c + 1] += cstart[c];
		std::vector<int64_t> fill(cstart.begin(), cstart.end() - 1);
		run_threads(nt, [&](int t) {
			for(int64_t e = ne * t / nt, e1 = ne * (t + 1) / nt; e < e1; ++ e)
				eorder[__atomic_fetch_add(&fill[key[e].first], (int64_t)1, __ATOMIC_RELAXED)] = e;
		});
		{
			std::vector<int64_t> w(nv + 1);
			for(int64_t c = 0; c <= nv; ++ c)
				w[c] = cstart[c] + c;
			balanced_cuts(w, nt, ccut);
		}
		run_threads(nt, [&](int t) {
		for(int64_t c = ccut[t]; c < ccut[t + 1]; ++ c) {
			const int64_t b = cstart[c], n = cstart[c + 1] - b;
			if(n <= 1)
				continue;
			if(n > 32) {
				std::sort(eorder.begin() + b, eorder.begin() + b + n, [&](int64_t x, int64_t y) {
					return key[x].second != key[y].second ? key[x].second < key[y].second : x < y; });
				continue;
			}
			for(int64_t i = 1; i < n; ++ i) { // insertion sort by (row, edge)
				const int64_t x = eorder[b + i], rx = key[x].second;
				int64_t j = i;
				for(; j > 0; -- j) {
					const int64_t y = eorder[b + j - 1], ry = key[y].second;
					if(ry < rx || (ry == rx && y < x))
						break;
					eorder[b + j] = y;
				}
				eorder[b + j] = x;
			}
		}
		});
	}
	clk.lap("edges sorted by block");
	Structure &st = ap->st;
	st.nb = nv;
	st.dim.assign(dim, dim + nv);
	st.base.resize(nv + 1);
	st.base[0] = 0;
	for(int64_t v = 0; v < nv; ++ v)
		st.base[v + 1] = st.base[v] + dim[v];
	st.n = st.base[nv];
	// Columns are independent once the edges are in block order: a first pass counts the blocks and values of every column
	// (ranges of columns on host threads), a serial prefix gives every column its place, a second pass writes.
	st.col_ptr.assign(nv + 1, 0);
	HVec<int64_t> v_doff(nv), c_ob(nv + 1, 0), c_off(nv + 1, 0); // per column: first off-diagonal block, first value
	run_threads(nt, [&](int t) {
		for(int64_t c = ccut[t]; c < ccut[t + 1]; ++ c) {
			int64_t nblk = 0, nval = 0, rprev = -1;
			for(int64_t q = cstart[c]; q < cstart[c + 1]; ++ q) {
				const int64_t r = key[eorder[q]].second;
				if(r != rprev) {
					++ nblk;
					nval += (int64_t)dim[r] * dim[c];
					rprev = r;
				}
			}
			c_ob[c + 1] = nblk;
			c_off[c + 1] = nval + (int64_t)dim[c] * dim[c];
		}
	});
	for(int64_t c = 0; c < nv; ++ c) {
		st.col_ptr[c + 1] = st.col_ptr[c] + c_ob[c + 1] + 1;
		c_ob[c + 1] += c_ob[c];
		c_off[c + 1] += c_off[c];
	}
	st.nnzb = st.col_ptr[nv];
	st.nvals = c_off[nv];
	const int64_t n_ob = c_ob[nv];
	HVec<int32_t> ob_ptr(n_ob + 1), ob_edge(ne);
	HVec<int64_t> ob_off(n_ob);
	ob_ptr[0] = 0;
	st.row_idx.resize(st.nnzb);
	st.blk_off.resize(st.nnzb);
	run_threads(nt, [&](int t) {
		for(int64_t c = ccut[t]; c < ccut[t + 1]; ++ c) {
			int64_t p = st.col_ptr[c], k = c_ob[c], off = c_off[c], q = cstart[c];
			const int64_t qe = cstart[c + 1];
			while(q < qe) {
				const int64_t r = key[eorder[q]].second;
				st.row_idx[p] = r;
				st.blk_off[p] = off;
				++ p;
				ob_off[k] = off;
				off += (int64_t)dim[r] * dim[c];
				for(; q < qe && key[eorder[q]].second == r; ++ q) {
					const int64_t e = eorder[q]; // stable sort: edges of one block stay in edge order
					ob_edge[q] = (int32_t)e | (v0[e] > v1[e] ? (int32_t)0x80000000 : 0);
				}
				ob_ptr[++ k] = (int32_t)q;
			}
			st.row_idx[p] = c;
			st.blk_off[p] = off;
			v_doff[c] = off;
		}
	});
	ap->n_ob = (int64_t)ob_off.size();
	clk.lap("Lambda structure");
	// ---- per-vertex contribution lists in edge order: (edge, side)
	HVec<int32_t> vl_ptr(nv + 1, 0), vl_entry(2 * ne);
	// (serial: the camera side of a BA graph is a few hundred counters that every edge increments -- atomic increments from
	// 16 threads on them measured 125 ms against 7 ms for this loop)
	for(int64_t e = 0; e < ne; ++ e) {
		++ vl_ptr[v0[e] + 1];
		++ vl_ptr[v1[e] + 1];
	}
	for(int64_t v = 0; v < nv; ++ v)
		vl_ptr[v + 1] += vl_ptr[v];
	{
		HVec<int32_t> fill(vl_ptr.begin(), vl_ptr.end() - 1);
		for(int64_t e = 0; e < ne; ++ e) { // within an edge vertex 0 registers before vertex 1
			vl_entry[fill[v0[e]] ++] = (int32_t)(e << 1);
			vl_entry[fill[v1[e]] ++] = (int32_t)(e << 1) | 1;
		}
	}
	for(int cls = 0; cls < 2; ++ cls) {
		ap->h_vlist_seq[cls].clear();
		ap->h_vlist_wave[cls].clear();
	}
	std::vector<int32_t> lseq[2], lwave[2];
	for(int64_t v = 0; v < nv; ++ v) {
		const int cls = (dim[v] == d0) ? 0 : 1;
		SPP_REQUIRE(dim[v] == d0 || dim[v] == d1, SPP_E_BADARG, "vertex width outside of the edge group");
		if(vl_ptr[v + 1] - vl_ptr[v] <= SEQ_MAX_DEGREE)
			lseq[cls].push_back((int32_t)v);
		else
			lwave[cls].push_back((int32_t)v);
	}
	clk.lap("vertex lists");
	hipStream_t s = ctx->stream;
	UploadArena arena(s);
	arena.add(ap->ob_ptr, ob_ptr);
	arena.add(ap->ob_edge, ob_edge);
	arena.add(ap->ob_off, ob_off);
	arena.add(ap->vl_ptr, vl_ptr);
	arena.add(ap->vl_entry, vl_entry);
	arena.add(ap->v_doff, v_doff);
	{
		std::vector<int64_t> vb(st.base.begin(), st.base.end() - 1);
		arena.add(ap->v_base, vb);
	}
	for(int cls = 0; cls < 2; ++ cls) {
		ap->n_seq[cls] = (int64_t)lseq[cls].size();
		ap->n_wave[cls] = (int64_t)lwave[cls].size();
		arena.add(ap->vlist_seq[cls], lseq[cls]);
		arena.add(ap->vlist_wave[cls], lwave[cls]);
	}
	arena.commit(ap->index_store);
	SPP_HIP_CHECK(hipStreamSynchronize(s));
	clk.lap("uploads");
	// complete: install. The ctx now describes this Lambda (sizes for spp_get_info before spp_analyze is called);
	// a solve plan analyzed for a DIFFERENT structure is dropped rather than left beside the new sizes
	if(ctx->mode >= 0 && (ctx->st.nb != st.nb || ctx->st.n != st.n || ctx->st.nnzb != st.nnzb || ctx->st.nvals != st.nvals))
		ctx->mode = -1;
	ctx->st.nb = st.nb; ctx->st.n = st.n; ctx->st.nnzb = st.nnzb; ctx->st.nvals = st.nvals;
	ctx->assemble = ap;
	guard.p = nullptr;
}

// --------------------------------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------------------------------
// T = J^T Omega  (D x RD), J is RD x D column-major
template <int D, int RD>
__device__ __forceinline__ void jt_omega(const double *__restrict__ J, const double *__restrict__ Om, double *T)
{
#pragma unroll
	for(int c = 0; c < RD; ++ c)
#pragma unroll
		for(int i = 0; i < D; ++ i) {
			double s = 0;
#pragma unroll
			for(int l = 0; l < RD; ++ l)
				s += J[l + i * RD] * Om[l + c * RD];
			T[i + c * D] = s;
		}
}

template <int D0, int D1, int RD>
__global__ __launch_bounds__(256)
void offdiag_kernel(int64_t n_ob, const int32_t *__restrict__ ob_ptr, const int32_t *__restrict__ ob_edge,
	const int64_t *__restrict__ ob_off, const double *__restrict__ J0, const double *__restrict__ J1,
	const double *__restrict__ Om, const double *__restrict__ wts, double *__restrict__ vals)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(b >= n_ob)
		return;
	double acc[D0 * D1];
	bool first = true;
	for(int32_t q = ob_ptr[b]; q < ob_ptr[b + 1]; ++ q) {
		const int32_t ee = ob_edge[q];
		const int64_t e = ee & 0x7fffffff;
		const bool rev = ee < 0;
		double j0[RD * D0], j1[RD * D1], om[RD * RD], T[D0 * RD];
#pragma unroll
		for(int i = 0; i < RD * D0; ++ i) j0[i] = J0[e * RD * D0 + i];
#pragma unroll
		for(int i = 0; i < RD * D1; ++ i) j1[i] = J1[e * RD * D1 + i];
#pragma unroll
		for(int i = 0; i < RD * RD; ++ i) om[i] = Om[e * RD * RD + i];
		jt_omega<D0, RD>(j0, om, T);
		if(wts) { // robust edge: t_H0_sigma_inv = J0^T Sigma^-1 w (BaseTypes_Binary.h:771)
			const double wgt = wts[e];
#pragma unroll
			for(int i = 0; i < D0 * RD; ++ i)
				T[i] *= wgt;
		}
		// H01 (D0 x D1) = T J1 ; stored as is, or transposed (D1 x D0) when the ids are reversed
#pragma unroll
		for(int c = 0; c < D1; ++ c)
#pragma unroll
			for(int i = 0; i < D0; ++ i) {
				double s = 0;
#pragma unroll
				for(int l = 0; l < RD; ++ l)
					s += T[i + l * D0] * j1[l + c * RD];
				const int idx = rev ? (c + i * D1) : (i + c * D0);
				acc[idx] = first ? s : acc[idx] + s;
			}
		first = false;
	}
	double *o = vals + ob_off[b];
#pragma unroll
	for(int i = 0; i < D0 * D1; ++ i)
		o[i] = acc[i];
}

// contribution of one (edge, side) to the vertex: H (D x D, upper computed, mirrored) and g (D)
template <int D, int RD, int SIDE>
__device__ __forceinline__ void vertex_contrib(const double *__restrict__ J, const double *__restrict__ Om,
	const double *__restrict__ r, double wgt, double *H, double *g)
{
	// wgt: the robust weight of the edge (1 for a plain edge: the products below are then exact), applied where the
	// reference applies it (BaseTypes_Binary.h:768-848): side 0 through T = J0^T Omega w -- H00 carries it once, g0 = T r w
	// TWICE --, side 1 on the finished H11 and g1
	double T[D * RD];
	jt_omega<D, RD>(J, Om, T);
	if(SIDE == 0) {
#pragma unroll
		for(int i = 0; i < D * RD; ++ i)
			T[i] *= wgt;
	}
#pragma unroll
	for(int c = 0; c < D; ++ c)
#pragma unroll
		for(int i = 0; i <= c; ++ i) {
			double s = 0;
#pragma unroll
			for(int l = 0; l < RD; ++ l)
				s += T[i + l * D] * J[l + c * RD];
			H[i + c * D] = (SIDE == 0) ? s : s * wgt;
		}
	if(SIDE == 0) { // g0 = (J0^T Omega) r
#pragma unroll
		for(int i = 0; i < D; ++ i) {
			double s = 0;
#pragma unroll
			for(int l = 0; l < RD; ++ l)
				s += T[i + l * D] * r[l];
			g[i] = s * wgt;
		}
	} else {        // g1 = J1^T (Omega r)
		double orr[RD];
#pragma unroll
		for(int l = 0; l < RD; ++ l) {
			double s = 0;
#pragma unroll
			for(int m = 0; m < RD; ++ m)
				s += Om[l + m * RD] * r[m];
			orr[l] = s;
		}
#pragma unroll
		for(int i = 0; i < D; ++ i) {
			double s = 0;
#pragma unroll
			for(int l = 0; l < RD; ++ l)
				s += J[l + i * RD] * orr[l];
			g[i] = s * wgt;
		}
	}
}

// D = width of the vertices handled; when D0 == D1 a vertex may sit on either side of its edges
template <int D, int D0, int D1, int RD>
__device__ __forceinline__ void load_contrib(int32_t entry, const double *__restrict__ J0, const double *__restrict__ J1,
	const double *__restrict__ Om, const double *__restrict__ r, const double *__restrict__ wts, double *H, double *g)
{
	const int64_t e = entry >> 1;
	const int side = entry & 1;
	const double wgt = wts ? wts[e] : 1.0;
	double om[RD * RD], rr[RD], j[RD * D];
#pragma unroll
	for(int i = 0; i < RD * RD; ++ i) om[i] = Om[e * RD * RD + i];
#pragma unroll
	for(int i = 0; i < RD; ++ i) rr[i] = r[e * RD + i];
	if(side == 0) {
		if(D == D0) {
#pragma unroll
			for(int i = 0; i < RD * D; ++ i) j[i] = J0[e * RD * D0 + i];
			vertex_contrib<D, RD, 0>(j, om, rr, wgt, H, g);
		}
	} else {
		if(D == D1) {
#pragma unroll
			for(int i = 0; i < RD * D; ++ i) j[i] = J1[e * RD * D1 + i];
			vertex_contrib<D, RD, 1>(j, om, rr, wgt, H, g);
		}
	}
}

template <int D>
__device__ __forceinline__ void store_vertex(const double *H, const double *g, bool unary, double damping,
	double *__restrict__ hd, double *__restrict__ gd)
{
#pragma unroll
	for(int c = 0; c < D; ++ c)
#pragma unroll
		for(int i = 0; i <= c; ++ i) {
			double v = H[i + c * D];
			if(i == c) {
				if(unary)
					v += 1.0;  // UF^T UF = identity, last in the reduction list (_Lambda_Base.h:1903-1924)
				v += damping;   // Lambda_ii.diagonal() += alpha (NonlinearSolver_Lambda_LM.h:228-239)
			}
			hd[i + c * D] = v;
			hd[c + i * D] = v;  // selfadjointView<Upper>: lower half mirrors the upper
		}
#pragma unroll
	for(int i = 0; i < D; ++ i)
		gd[i] = g[i];
}

template <int D, int D0, int D1, int RD>
__global__ __launch_bounds__(256)
void vertex_seq_kernel(int64_t nlist, const int32_t *__restrict__ vlist, const int32_t *__restrict__ vl_ptr,
	const int32_t *__restrict__ vl_entry, const int64_t *__restrict__ v_doff, const int64_t *__restrict__ v_base,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Om,
	const double *__restrict__ r, const double *__restrict__ wts, int64_t unary_vertex, double damping, double *__restrict__ vals,
	double *__restrict__ eta)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(t >= nlist)
		return;
	const int32_t v = vlist[t];
	double H[D * D], g[D];
#pragma unroll
	for(int i = 0; i < D * D; ++ i) H[i] = 0;
#pragma unroll
	for(int i = 0; i < D; ++ i) g[i] = 0;
	bool first = true;
	for(int32_t q = vl_ptr[v]; q < vl_ptr[v + 1]; ++ q) {
		double Hc[D * D], gc[D];
		load_contrib<D, D0, D1, RD>(vl_entry[q], J0, J1, Om, r, wts, Hc, gc);
		if(first) { // the first source is assigned, the others are added (_Lambda_Base.h:598-604)
#pragma unroll
			for(int c = 0; c < D; ++ c)
#pragma unroll
				for(int i = 0; i <= c; ++ i) H[i + c * D] = Hc[i + c * D];
#pragma unroll
			for(int i = 0; i < D; ++ i) g[i] = gc[i];
			first = false;
		} else {
#pragma unroll
			for(int c = 0; c < D; ++ c)
#pragma unroll
				for(int i = 0; i <= c; ++ i) H[i + c * D] += Hc[i + c * D];
#pragma unroll
			for(int i = 0; i < D; ++ i) g[i] += gc[i];
		}
	}
	store_vertex<D>(H, g, v == unary_vertex, damping, vals + v_doff[v], eta + v_base[v]);
}

template <int D, int D0, int D1, int RD>
__global__ __launch_bounds__(256)
void vertex_wave_kernel(int64_t nlist, const int32_t *__restrict__ vlist, const int32_t *__restrict__ vl_ptr,
	const int32_t *__restrict__ vl_entry, const int64_t *__restrict__ v_doff, const int64_t *__restrict__ v_base,
	const double *__restrict__ J0, const double *__restrict__ J1, const double *__restrict__ Om,
	const double *__restrict__ r, const double *__restrict__ wts, int64_t unary_vertex, double damping, double *__restrict__ vals,
	double *__restrict__ eta)
{
	const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if(t >= nlist)
		return;
	const int32_t v = vlist[t];
	double H[D * D], g[D];
#pragma unroll
	for(int i = 0; i < D * D; ++ i) H[i] = 0;
#pragma unroll
	for(int i = 0; i < D; ++ i) g[i] = 0;
	for(int32_t q = vl_ptr[v] + lane; q < vl_ptr[v + 1]; q += 64) {
		double Hc[D * D], gc[D];
		load_contrib<D, D0, D1, RD>(vl_entry[q], J0, J1, Om, r, wts, Hc, gc);
#pragma unroll
		for(int c = 0; c < D; ++ c)
#pragma unroll
			for(int i = 0; i <= c; ++ i) H[i + c * D] += Hc[i + c * D];
#pragma unroll
		for(int i = 0; i < D; ++ i) g[i] += gc[i];
	}
#pragma unroll
	for(int c = 0; c < D; ++ c)
#pragma unroll
		for(int i = 0; i <= c; ++ i) {
			double x = H[i + c * D];
#pragma unroll
			for(int off = 32; off > 0; off >>= 1)
				x += __shfl_xor(x, off);
			H[i + c * D] = x;
		}
#pragma unroll
	for(int i = 0; i < D; ++ i) {
		double x = g[i];
#pragma unroll
		for(int off = 32; off > 0; off >>= 1)
			x += __shfl_xor(x, off);
		g[i] = x;
	}
	if(lane == 0)
		store_vertex<D>(H, g, v == unary_vertex, damping, vals + v_doff[v], eta + v_base[v]);
}

template <int D0, int D1, int RD>
static void assemble_t(spp_ctx *ctx, const double *J0, const double *J1, const double *Om, const double *r,
	double damping, double *vals, double *eta)
{
	AssemblePlan *ap = ctx->assemble;
	hipStream_t s = ctx->stream;
	if(ap->n_ob)
		hipLaunchKernelGGL((offdiag_kernel<D0, D1, RD>), dim3((unsigned)((ap->n_ob + 255) / 256)), dim3(256), 0, s,
			ap->n_ob, ap->ob_ptr.p, ap->ob_edge.p, ap->ob_off.p, J0, J1, Om, ap->edge_weights, vals);
#define SPP_VERTEX_LAUNCH(D, cls) \
	if(ap->n_seq[cls]) \
		hipLaunchKernelGGL((vertex_seq_kernel<D, D0, D1, RD>), dim3((unsigned)((ap->n_seq[cls] + 255) / 256)), dim3(256), 0, s, \
			ap->n_seq[cls], ap->vlist_seq[cls].p, ap->vl_ptr.p, ap->vl_entry.p, ap->v_doff.p, ap->v_base.p, \
			J0, J1, Om, r, ap->edge_weights, ap->unary_vertex, damping, vals, eta); \
	if(ap->n_wave[cls]) \
		hipLaunchKernelGGL((vertex_wave_kernel<D, D0, D1, RD>), dim3((unsigned)((ap->n_wave[cls] + 3) / 4)), dim3(256), 0, s, \
			ap->n_wave[cls], ap->vlist_wave[cls].p, ap->vl_ptr.p, ap->vl_entry.p, ap->v_doff.p, ap->v_base.p, \
			J0, J1, Om, r, ap->edge_weights, ap->unary_vertex, damping, vals, eta);
	SPP_VERTEX_LAUNCH(D0, 0)
	if(D0 != D1) {
		SPP_VERTEX_LAUNCH(D1, 1)
	}
#undef SPP_VERTEX_LAUNCH
	SPP_HIP_CHECK(hipGetLastError());
}

void assemble_run(spp_ctx *ctx, const double *J0, const double *J1, const double *Om, const double *r,
	double damping, double *vals, double *eta)
{
	AssemblePlan *ap = ctx->assemble;
	if(ap->d0 == 6 && ap->d1 == 3) assemble_t<6, 3, 2>(ctx, J0, J1, Om, r, damping, vals, eta);
	else if(ap->d0 == 3 && ap->d1 == 3) assemble_t<3, 3, 3>(ctx, J0, J1, Om, r, damping, vals, eta);
	else if(ap->d0 == 6 && ap->d1 == 6) assemble_t<6, 6, 6>(ctx, J0, J1, Om, r, damping, vals, eta);
	else assemble_t<3, 2, 2>(ctx, J0, J1, Om, r, damping, vals, eta);
}

} // namespace spp
