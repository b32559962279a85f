// spp_symbolic.cpp -- host-side symbolic analysis (integer work, once per block structure).
//
//  * schur_applicable / build_schur_plan: the guided Schur ordering and everything the reference
//    recomputes structurally in every CLinearSolver_Schur::Solve_PosDef_Blocky call
//      n_Calculate_GuidedOrdering      include/slam/LinearSolver_Schur.h:2154-2207, src/slam/LinearSolver_Schur.cpp:771-838
//      Permute_UpperTriangular_To      src/slam/BlockMatrix.cpp:8183 (replaced by index indirection: no data moves)
//      SliceTo x3 + TransposeTo        include/slam/LinearSolver_Schur.h:1699-1709 (replaced by the obs / pair lists)
//      symbolic part of MultiplyToWith src/slam/BlockMatrixFBS.inl:1147-1304 (the S block pattern + pair lists)
//  * min_degree_order: fill-reducing ordering of the block graph (the role of
//    CMatrixOrdering::p_BlockOrdering, src/slam/OrderingMagic.cpp:701-1034, which calls amd_l2).
//    This is our own quotient-graph approximate-minimum-degree implementation written from the
//    published algorithm (Amestoy, Davis, Duff 1996); it is NOT the reference's AMD code and gives a
//    different (equally valid) elimination order. Delta-x parity does not depend on the order.

#include "spp_internal.h"
#include <sys/mman.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <thread>
#include <memory>
#include <chrono>
#include <exception>

namespace spp {

void SchurPlan::release_all()
{
	lm_ptr.release(); bs_ptr.release(); n_bs = 0; lm_coff.release(); lm_rbase.release(); obs_pose.release(); obs_lm.release();
	obs_off.release(); pose_rbase.release(); cam_ptr.release(); cam_obs.release(); items.release();
	obs_wpos.release(); xcd_beg.release(); sblk_i1.release(); sblk_i2.release(); sblk_aoff.release();
	sblk_voff.release(); s_st = Structure(); sparse_S = false; mis = false;
	pair_a.release(); pair_b.release(); multi_blk.release(); multi_ptr.release(); cinv.release(); lfac.release();
	W.release(); Up.release(); xw.release(); partial.release(); S.release();
	pose_block.clear(); lm_block.clear(); is_lm.clear();
}

// Guided ordering is possible when there are exactly two block widths and the blocks of the
// smaller width (landmarks) are not connected to each other (C block diagonal). Mirrors
// LinearSolver_Schur.h:1586-1594 (fall back when there are not two vertex dimensions) and
// :1721-1726 (the fast path requires b_BlockDiagonal()).
bool schur_applicable(const Structure &st, int *dp_out, int *dl_out)
{
	int d_a = -1, d_b = -1;
	for(int64_t j = 0; j < st.nb; ++ j) {
		int d = st.dim[j];
		if(d_a < 0 || d == d_a) d_a = d;
		else if(d_b < 0 || d == d_b) d_b = d;
		else return false;
	}
	if(d_a < 0 || d_b < 0)
		return false;
	int dp = std::max(d_a, d_b), dl = std::min(d_a, d_b);
	if(!((dp == 6 && dl == 3) || (dp == 3 && dl == 2)))
		return false; // kernel instantiations (BA: SE3 pose + XYZ; 2D SLAM: SE2 pose + XY)
	int64_t n_lm = 0;
	for(int64_t j = 0; j < st.nb; ++ j) {
		if(st.dim[j] == dl)
			++ n_lm;
		for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
			int64_t i = st.row_idx[p];
			if(i != j && st.dim[i] == dl && st.dim[j] == dl)
				return false; // landmark-landmark block: C not block diagonal
		}
	}
	if(n_lm == 0)
		return false;
	*dp_out = dp;
	*dl_out = dl;
	return true;
}

static const int PAIR_CHUNK = 2048; // pairs per work item of the S accumulation

int64_t schur_buffer_doubles(const spp_ctx *ctx)
{
	const SchurPlan &sp = ctx->schur;
	return sp.sparse_S ? sp.s_st.nvals + sp.n_red : sp.ld * sp.ld;
}

// Maximum-independent-set cut for graphs of ONE block width (the general ordering of the reference,
// CSchurOrdering, src/slam/LinearSolver_Schur.cpp:690-769,1235-1340): vertices of the independent set play the
// landmarks' role (their diagonal part C is block diagonal by construction), the rest forms the reduced
// system. Greedy by ascending degree (ties by index): deterministic, maximal, not maximum.
static bool mis_partition(const Structure &st, std::vector<uint8_t> &is_lm, int *d_out)
{
	const int d = st.dim[0];
	for(int64_t j = 0; j < st.nb; ++ j)
		if(st.dim[j] != d)
			return false;
	if(d != 3 && d != 6)
		return false;
	std::vector<std::vector<int32_t> > adj(st.nb);
	for(int64_t j = 0; j < st.nb; ++ j)
		for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
			const int64_t i = st.row_idx[p];
			if(i != j) {
				adj[i].push_back((int32_t)j);
				adj[j].push_back((int32_t)i);
			}
		}
	std::vector<int32_t> order(st.nb);
	for(int64_t j = 0; j < st.nb; ++ j)
		order[j] = (int32_t)j;
	std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return adj[a].size() < adj[b].size(); });
	is_lm.assign(st.nb, 0);
	std::vector<uint8_t> blocked(st.nb, 0);
	int64_t n_lm = 0;
	for(int64_t q = 0; q < st.nb; ++ q) {
		const int32_t v = order[q];
		if(blocked[v])
			continue;
		is_lm[v] = 1;
		++ n_lm;
		for(size_t e = 0; e < adj[v].size(); ++ e)
			blocked[adj[v][e]] = 1;
	}
	*d_out = d;
	return n_lm > 0 && n_lm < st.nb;
}

// uninitialized host array (a std::vector would zero-fill -- and page-fault -- 100 MB on one thread)
template <class T>
struct RawBuf {
	T *p = nullptr;
	size_t n = 0;
	RawBuf() {}
	RawBuf(const RawBuf&) = delete;
	RawBuf &operator=(const RawBuf&) = delete;
	~RawBuf() { free(p); }
	void resize(size_t m)
	{
		free(p);
		p = nullptr;
		if(m) {
			// big work arrays (tens of MB, written once front to back): 2 MB-aligned and offered to transparent huge pages --
			// first touch of 260 MB in 4 KB pages is 63 000 page faults, a third of the pair-list phase
			const size_t bytes = m * sizeof(T);
			if(bytes >= ((size_t)8 << 20)) {
				void *q = nullptr;
				if(posix_memalign(&q, (size_t)2 << 20, (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1)) == 0) {
					p = (T*)q;
#ifdef MADV_HUGEPAGE
					madvise(q, bytes, MADV_HUGEPAGE);
#endif
				}
			}
			if(!p)
				p = (T*)malloc(bytes);
			if(!p)
				throw std::bad_alloc();
		}
		n = m;
	}
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; }
	const T &operator[](size_t i) const { return p[i]; }
};

// Everything build_schur_plan() derives from the block structure, in host memory: the pure symbolic part (no device,
// no ctx), also reachable through spp_schur_plan_host() for host-only tests and timing.
struct SchurPlanHost {
	int dp = 0, dl = 0;
	int64_t nc = 0, nl = 0, nl_total = 0, no = 0, n_red = 0, ld = 0, n_sblk = 0, n_pairs = 0, n_items = 0, n_multi = 0, n_ablk = 0;
	int32_t n_slots = 0, xcd_max_items = 0;
	bool u_landmark_major = true, factored = true;
	std::vector<int64_t> pose_block, lm_block;
	std::vector<uint8_t> is_lm;
	HVec<int32_t> lm_ptr, obs_pose, obs_lm, cam_obs, wpos; // (per observation / landmark: HVec = not zero-filled, huge pages)
	std::vector<int32_t> bs_ptr; // landmark ranges of at most BS_OBS observations (fused back-substitution); empty: a landmark has more
	std::vector<int32_t> cam_ptr, sblk_i1, sblk_i2, multi_blk, multi_ptr, xb;
	RawBuf<int32_t> pair_a, pair_b; // (tens of millions of entries: not value-initialized, first touched by the threads that fill them)
	HVec<int64_t> lm_coff, obs_off, lm_rbase;
	std::vector<int64_t> sblk_aoff, sblk_voff, pose_rbase;
	std::vector<SaccItem> recs;
	Structure s_st;
};

static void schur_plan_host(const Structure &st, int shard_rank, int shard_world, bool sparse_S, bool mis, SchurPlanHost &h)
{
	VClock clk("schur plan");
	int dp, dl;
	std::vector<uint8_t> &is_lm = h.is_lm;
	if(mis) {
		SPP_REQUIRE(mis_partition(st, is_lm, &dp), SPP_E_UNSUPPORTED,
			"MIS Schur mode needs a graph of one block width (3 or 6) with at least one edge");
		dl = dp;
	} else {
		SPP_REQUIRE(schur_applicable(st, &dp, &dl), SPP_E_UNSUPPORTED,
			"Schur mode needs exactly two block widths ({6,3} or {3,2}) and a block-diagonal landmark part");
		is_lm.assign(st.nb, 0);
		for(int64_t j = 0; j < st.nb; ++ j)
			is_lm[j] = st.dim[j] == dl;
	}
	h.dp = dp;
	h.dl = dl;

	// ---- guided ordering: stable partition by width (LinearSolver_Schur.cpp:771-838)
	std::vector<int32_t> pose_of(st.nb, -1), lm_of(st.nb, -1);
	int64_t nc = 0, nl_total = 0, nl = 0;
	for(int64_t j = 0; j < st.nb; ++ j) {
		if(!is_lm[j]) {
			pose_of[j] = (int32_t)nc ++;
			h.pose_block.push_back(j);
		} else {
			// landmark sharding: round-robin over ranks keeps track lengths balanced
			if(nl_total % shard_world == shard_rank) {
				lm_of[j] = (int32_t)nl ++;
				h.lm_block.push_back(j);
			}
			++ nl_total;
		}
	}
	h.nc = nc;
	h.nl = nl;
	h.nl_total = nl_total;
	const bool add_A = (shard_rank == 0);
	h.n_red = nc * dp;
	h.ld = ((h.n_red + 1 + DENSE_NB - 1) / DENSE_NB) * DENSE_NB; // at least one padding column (rhs)
	SPP_REQUIRE(sparse_S || h.ld <= 65536, SPP_E_UNSUPPORTED,
		"reduced camera system too large for the dense path (use SPP_MODE_SCHUR_SPARSE)");

	// ---- observations: every pose-landmark block, sorted by (landmark, pose). Two passes over ranges of columns on host
	// threads: counts per range, then every range writes its observations / camera-camera blocks at its offset.
	struct Obs { int32_t lm, pose; int64_t off; };
	HVec<int64_t> &lm_coff = h.lm_coff;
	lm_coff.assign(nl, -1);
	struct ABlk { int32_t i1, i2; int64_t off; };
	std::vector<ABlk> ablk;
	HVec<int32_t> &lm_ptr = h.lm_ptr, &obs_pose = h.obs_pose, &obs_lm = h.obs_lm;
	HVec<int64_t> &obs_off = h.obs_off;
	bool obs_sorted = true;
	const int nts = plan_threads(st.nnzb);
	std::vector<int64_t> jcut;
	balanced_cuts(st.col_ptr, nts, jcut);
	{
		std::vector<int64_t> n_obs_t(nts + 1, 0), n_ab_t(nts + 1, 0);
		// what block p of column j is: 0 camera-camera, 1 landmark diagonal, 2 observation (o filled), 3 not of this shard
		auto classify = [&](int64_t j, int64_t p, Obs &o) -> int {
			const int64_t i = st.row_idx[p]; // i <= j
			const bool pi = !is_lm[i], pj = !is_lm[j];
			if(pi && pj)
				return 0;
			if(!pi && !pj)
				return 1;
			if(pi) { // block (pose i, landmark j): dp x dl as stored
				if(lm_of[j] < 0)
					return 3;
				o = {lm_of[j], pose_of[i], st.blk_off[p] << 1};
			} else { // block (landmark i, pose j): stored transposed, dl x dp
				if(lm_of[i] < 0)
					return 3;
				o = {lm_of[i], pose_of[j], (st.blk_off[p] << 1) | 1};
			}
			return 2;
		};
		run_threads(nts, [&](int t) {
			int64_t n_o = 0, n_a = 0;
			Obs o;
			for(int64_t j = jcut[t]; j < jcut[t + 1]; ++ j)
				for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
					const int k = classify(j, p, o);
					n_o += k == 2;
					n_a += k == 0;
				}
			n_obs_t[t + 1] = n_o;
			n_ab_t[t + 1] = n_a;
		});
		for(int t = 0; t < nts; ++ t) {
			n_obs_t[t + 1] += n_obs_t[t];
			n_ab_t[t + 1] += n_ab_t[t];
		}
		const int64_t no_all = n_obs_t[nts];
		SPP_REQUIRE(no_all < (int64_t(1) << 31), SPP_E_UNSUPPORTED, "too many observations for 32-bit obs indices");
		obs_pose.resize(no_all);
		obs_lm.resize(no_all);
		obs_off.resize(no_all);
		ablk.resize(n_ab_t[nts]);
		std::vector<char> sorted_t(nts, 1);
		run_threads(nts, [&](int t) {
			int64_t a = n_obs_t[t], q = n_ab_t[t];
			Obs o, prev = {-1, -1, 0};
			bool sorted = true;
			for(int64_t j = jcut[t]; j < jcut[t + 1]; ++ j)
				for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
					const int k = classify(j, p, o);
					if(k == 0)
						ablk[q ++] = {pose_of[st.row_idx[p]], pose_of[j], st.blk_off[p]}; // i <= j and stable partition keep i1 <= i2
					else if(k == 1) {
						if(lm_of[j] >= 0)
							lm_coff[lm_of[j]] = st.blk_off[p]; // diagonal C block
					} else if(k == 2) {
						if(o.lm < prev.lm || (o.lm == prev.lm && o.pose < prev.pose))
							sorted = false;
						prev = o;
						obs_lm[a] = o.lm;
						obs_pose[a] = o.pose;
						obs_off[a] = o.off;
						++ a;
					}
				}
			sorted_t[t] = sorted;
		});
		for(int t = 0; t < nts; ++ t)
			obs_sorted = obs_sorted && sorted_t[t];
		for(int t = 1; t < nts && obs_sorted; ++ t) { // across the ranges
			const int64_t a = n_obs_t[t];
			if(a > 0 && a < no_all && (obs_lm[a] < obs_lm[a - 1] || (obs_lm[a] == obs_lm[a - 1] && obs_pose[a] < obs_pose[a - 1])))
				obs_sorted = false;
		}
	}
	h.n_ablk = (int64_t)ablk.size();
	clk.lap("partition + observation scan");
	for(int64_t l = 0; l < nl; ++ l)
		SPP_REQUIRE(lm_coff[l] >= 0, SPP_E_BADARG, "landmark without a diagonal block");
	// Sharded + sparse reduced system: every rank must hold the SAME block structure of S (the union
	// over all landmarks), or the all-reduce of the value arrays would add unrelated blocks. The pattern
	// of the landmarks this rank does not own is collected here (pose lists per foreign landmark).
	std::vector<std::vector<int32_t> > foreign_cols; // per pose i1: poses i2 > i1 co-observing a foreign landmark
	if(sparse_S && shard_world > 1) {
		std::vector<int64_t> lm_gidx(st.nb, -1);
		int64_t g = 0;
		for(int64_t j = 0; j < st.nb; ++ j)
			if(is_lm[j])
				lm_gidx[j] = g ++;
		std::vector<std::vector<int32_t> > poses_of(g);
		for(int64_t j = 0; j < st.nb; ++ j)
			for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
				const int64_t i = st.row_idx[p];
				const bool pi = !is_lm[i], pj = !is_lm[j];
				if(pi && !pj && lm_of[j] < 0)
					poses_of[lm_gidx[j]].push_back(pose_of[i]);
				else if(!pi && pj && lm_of[i] < 0)
					poses_of[lm_gidx[i]].push_back(pose_of[j]);
			}
		foreign_cols.resize(nc);
		for(int64_t l = 0; l < g; ++ l) {
			std::vector<int32_t> &ps = poses_of[l];
			std::sort(ps.begin(), ps.end());
			for(size_t a = 0; a < ps.size(); ++ a)
				for(size_t b = a + 1; b < ps.size(); ++ b)
					foreign_cols[ps[a]].push_back(ps[b]);
		}
		for(int64_t c = 0; c < nc; ++ c) {
			std::sort(foreign_cols[c].begin(), foreign_cols[c].end());
			foreign_cols[c].erase(std::unique(foreign_cols[c].begin(), foreign_cols[c].end()), foreign_cols[c].end());
		}
	}
	// (cameras before points, the usual numbering: the column scan above already emits the observations in order)
	const int64_t no = (int64_t)obs_pose.size();
	h.no = no;
	if(!obs_sorted) {
		std::vector<Obs> obs(no);
		for(int64_t a = 0; a < no; ++ a)
			obs[a] = {obs_lm[a], obs_pose[a], obs_off[a]};
		std::sort(obs.begin(), obs.end(), [](const Obs &a, const Obs &b) {
			return a.lm != b.lm ? a.lm < b.lm : a.pose < b.pose; });
		for(int64_t a = 0; a < no; ++ a) {
			obs_lm[a] = obs[a].lm;
			obs_pose[a] = obs[a].pose;
			obs_off[a] = obs[a].off;
		}
	}
	// lm_ptr[l] = number of observations of the landmarks before l: written where the landmark changes
	lm_ptr.resize(nl + 1);
	const int nto = plan_threads(no);
	run_threads(nto, [&](int t) {
		const int64_t a0 = no * t / nto, a1 = no * (t + 1) / nto;
		for(int64_t a = a0; a < a1; ++ a) {
			const int32_t lp = a ? obs_lm[a - 1] : -1, lc = obs_lm[a];
			for(int32_t l = lp + 1; l <= lc; ++ l)
				lm_ptr[l] = (int32_t)a;
		}
	});
	for(int64_t l = (no ? obs_lm[no - 1] : -1) + 1; l <= nl; ++ l)
		lm_ptr[l] = (int32_t)no;

	// ---- back-substitution: consecutive landmarks in groups of at most 256 observations (one workgroup each: the products
	// U^T dx of a group stay in LDS, spp_schur.hip backsubst_fused_kernel)
	{
		const int32_t BS_OBS = 256;
		std::vector<int32_t> &bp = h.bs_ptr;
		bp.clear();
		bp.reserve((size_t)(no / 160 + 2));
		bp.push_back(0);
		int32_t first = 0;
		bool ok = true;
		for(int64_t l = 0; l < nl; ++ l) {
			if(lm_ptr[l + 1] - lm_ptr[l] > BS_OBS) {
				ok = false;
				break;
			}
			if(lm_ptr[l + 1] - lm_ptr[first] > BS_OBS || l - first >= BS_OBS) { // (one lane per observation, then one per landmark)
				bp.push_back((int32_t)l);
				first = (int32_t)l;
			}
		}
		if(ok && nl)
			bp.push_back((int32_t)nl);
		else
			bp.clear();
	}

	// ---- per-pose observation lists (ascending landmark = ascending obs index): a counting sort by camera, ranges of
	// observations on host threads (per-range, per-camera counts give every range its place in every camera's list)
	std::vector<int32_t> &cam_ptr = h.cam_ptr;
	HVec<int32_t> &cam_obs = h.cam_obs;
	cam_ptr.assign(nc + 1, 0);
	cam_obs.resize(no);
	// (beside every entry: where the observations of its landmark end -- the pair lists below walk [a, cam_end) per entry and
	// would otherwise chase cam_obs -> obs_lm -> lm_ptr, three dependent cache misses, per observation)
	HVec<int32_t> cam_end(no);
	{
		std::vector<int32_t> cnt((size_t)nto * nc, 0); // [range][camera]
		run_threads(nto, [&](int t) {
			int32_t *c = cnt.data() + (size_t)t * nc;
			for(int64_t a = no * t / nto, a1 = no * (t + 1) / nto; a < a1; ++ a)
				++ c[obs_pose[a]];
		});
		for(int64_t c = 0; c < nc; ++ c) {
			int32_t sum = cam_ptr[c];
			for(int t = 0; t < nto; ++ t) {
				const int32_t v = cnt[(size_t)t * nc + c];
				cnt[(size_t)t * nc + c] = sum; // where range t starts in the list of camera c
				sum += v;
			}
			cam_ptr[c + 1] = sum;
		}
		run_threads(nto, [&](int t) {
			int32_t *fill = cnt.data() + (size_t)t * nc;
			for(int64_t a = no * t / nto, a1 = no * (t + 1) / nto; a < a1; ++ a) {
				const int32_t pos = fill[obs_pose[a]] ++;
				cam_obs[pos] = (int32_t)a;
				cam_end[pos] = lm_ptr[obs_lm[a] + 1];
			}
		});
	}

	// camera-major position of every observation: W, Up, xw are stored in this order, so that the
	// blocks one camera contributes are contiguous (the S accumulation gathers them per camera pair)
	HVec<int32_t> &wpos = h.wpos;
	wpos.resize(no);
	run_threads(nto, [&](int t) {
		for(int64_t q = no * t / nto, q1 = no * (t + 1) / nto; q < q1; ++ q)
			wpos[cam_obs[q]] = (int32_t)q;
	});

	clk.lap("observation / camera lists");
	// ---- S block pattern and pair lists. Key = (i1 <= i2). The pairs of a block keep the landmark order, which is the
	// reference's accumulation order (MultiplyToWith_FBS walks the columns of V = landmarks in ascending order).
	// Built row by row of S on host threads (ranges of rows balanced by their pair counts).
	std::vector<int64_t> lm_pairs(nl + 1, 0);
	for(int64_t l = 0; l < nl; ++ l) {
		const int64_t k = lm_ptr[l + 1] - lm_ptr[l];
		lm_pairs[l + 1] = lm_pairs[l] + k * (k + 1) / 2;
	}
	const int64_t n_pairs = lm_pairs[nl];
	SPP_REQUIRE(n_pairs < (int64_t(1) << 31), SPP_E_UNSUPPORTED, "too many block products for 32-bit pair indices");
	h.n_pairs = n_pairs;
	{
		const char *e = getenv("SPP_SACC_ULM");
		h.u_landmark_major = e ? atoi(e) != 0 : true;
		e = getenv("SPP_SACC_FACTORED"); // 0: two packed blocks per observation (W and U), as in rounds 1-2
		h.factored = e ? atoi(e) != 0 : true;
	}
	RawBuf<int32_t> &pair_a = h.pair_a, &pair_b = h.pair_b;
	pair_a.resize(n_pairs);
	pair_b.resize(n_pairs);
	{
		// first touch in parallel, in order: the fill below writes ~200 000 interleaved streams, and page faults taken in
		// that order by many threads at once serialize in the kernel
		const int ntt = plan_threads(n_pairs);
		run_threads(ntt, [&](int t) {
			const int64_t b = n_pairs * t / ntt, e = n_pairs * (t + 1) / ntt;
			memset(pair_a.p + b, 0, (size_t)(e - b) * sizeof(int32_t));
			memset(pair_b.p + b, 0, (size_t)(e - b) * sizeof(int32_t));
		});
	}
	clk.lap("pair list buffers");
	std::vector<int32_t> &sblk_i1 = h.sblk_i1, &sblk_i2 = h.sblk_i2;
	std::vector<int64_t> &sblk_aoff = h.sblk_aoff;
	std::vector<int64_t> sblk_beg; // pair range per S block
	const int nt = plan_threads(n_pairs);
	const bool ulm = h.u_landmark_major, fact = h.factored;
	// A blocks grouped by row for merging
	std::vector<std::vector<std::pair<int32_t, int64_t> > > a_by_row(nc);
	for(size_t q = 0; q < ablk.size(); ++ q)
		a_by_row[ablk[q].i1].push_back(std::make_pair(ablk[q].i2, ablk[q].off));
	{
		// Row-driven: row i1 of S enumerates its pairs itself -- the observations a of camera i1 (ascending landmark), and
		// for each the observations b >= a of the same landmark (ascending pose = column i2 >= i1) -- once to count its
		// columns and once to write the lists. Nothing is bucketed through memory: the pairs of a row go straight to their
		// place, the observer lists they are enumerated from (11 MB on the Venice shape) stay in cache.
		std::vector<int64_t> row_cnt(nc + 1, 0);
		{
			std::vector<int64_t> ccut;
			balanced_cuts(std::vector<int64_t>(cam_ptr.begin(), cam_ptr.end()), nt, ccut);
			run_threads(nt, [&](int t) {
				for(int64_t c = ccut[t]; c < ccut[t + 1]; ++ c) {
					int64_t sum = 0;
					for(int32_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++ q)
						sum += cam_end[q] - cam_obs[q]; // pairs (a, b >= a)
					row_cnt[c + 1] = sum;
				}
			});
			for(int64_t c = 0; c < nc; ++ c)
				row_cnt[c + 1] += row_cnt[c];
		}
		clk.lap("pair counts");
		// pass 2: rows are independent (row i1 writes the pairs [row_cnt[i1], row_cnt[i1 + 1]))
		std::vector<int64_t> rcut;
		{
			std::vector<int64_t> w(nc + 1, 0);
			for(int64_t c = 0; c < nc; ++ c)
				w[c + 1] = w[c] + (row_cnt[c + 1] - row_cnt[c]) + (nc - c); // pairs + the scan over the row's columns
			balanced_cuts(w, nt, rcut);
		}
		struct RowOut { std::vector<int32_t> i1, i2; std::vector<int64_t> aoff, beg; };
		std::vector<RowOut> rout(nt);
		run_threads(nt, [&](int t) {
			RowOut &ro = rout[t];
			std::vector<int64_t> col_cnt(nc + 1), a_of_col(nc), start(nc + 1, 0);
			std::vector<char> foreign(nc, 0);
			for(int64_t i1 = rcut[t]; i1 < rcut[t + 1]; ++ i1) {
				const int64_t b0 = row_cnt[i1], b1 = row_cnt[i1 + 1];
				std::fill(col_cnt.begin() + i1, col_cnt.end(), 0);
				std::fill(a_of_col.begin() + i1, a_of_col.end(), -1);
				const int32_t qa0 = cam_ptr[i1], qa1 = cam_ptr[i1 + 1];
				constexpr int32_t AHEAD = 24; // (the observers of the entries ahead: the only access that is not a stream)
				for(int32_t q = qa0; q < qa1; ++ q) {
					if(q + AHEAD < (int32_t)no)
						__builtin_prefetch(&obs_pose[cam_obs[q + AHEAD]]);
					const int32_t a = cam_obs[q], e = cam_end[q];
					for(int32_t b = a; b < e; ++ b)
						++ col_cnt[obs_pose[b] + 1];
				}
				for(size_t q = 0; q < a_by_row[i1].size(); ++ q)
					a_of_col[a_by_row[i1][q].first] = a_by_row[i1][q].second;
				if(!foreign_cols.empty())
					for(size_t q = 0; q < foreign_cols[i1].size(); ++ q)
						foreign[foreign_cols[i1][q]] = 1;
				// blocks of this row, ascending i2
				int64_t out = b0;
				for(int64_t c = i1; c < nc; ++ c) {
					start[c] = out;
					if(col_cnt[c + 1] || a_of_col[c] >= 0 || foreign[c]) {
						ro.i1.push_back((int32_t)i1);
						ro.i2.push_back((int32_t)c);
						ro.aoff.push_back(a_of_col[c]);
						ro.beg.push_back(out);
					}
					out += col_cnt[c + 1];
				}
				if(!foreign_cols.empty())
					for(size_t q = 0; q < foreign_cols[i1].size(); ++ q)
						foreign[foreign_cols[i1][q]] = 0;
				// stable: landmark order preserved. The pair lists address W / Up, i.e. camera-major positions (the packed
				// U either camera-major like W, or landmark-major = in observation order: the blocks of one landmark's
				// observers are then one contiguous run, which the blocks of one ROW of S gather together)
				for(int32_t q = qa0; q < qa1; ++ q) {
					if(q + AHEAD < (int32_t)no)
						__builtin_prefetch(&obs_pose[cam_obs[q + AHEAD]]);
					const int32_t a = cam_obs[q], e = cam_end[q];
					const int32_t wa = fact ? a : q; // (the position the pair lists address: observation order in the factored form, else camera-major = wpos[a])
					for(int32_t b = a; b < e; ++ b) {
						int64_t &f = start[obs_pose[b]];
						pair_a[f] = wa;
						pair_b[f] = (ulm || fact) ? b : wpos[b];
						++ f;
					}
				}
				(void)b1;
			}
		});
		clk.lap("pair lists by row");
		size_t nblk = 0;
		for(int t = 0; t < nt; ++ t)
			nblk += rout[t].i1.size();
		sblk_i1.reserve(nblk);
		sblk_i2.reserve(nblk);
		sblk_aoff.reserve(nblk);
		sblk_beg.reserve(nblk + 1);
		for(int t = 0; t < nt; ++ t) {
			sblk_i1.insert(sblk_i1.end(), rout[t].i1.begin(), rout[t].i1.end());
			sblk_i2.insert(sblk_i2.end(), rout[t].i2.begin(), rout[t].i2.end());
			sblk_aoff.insert(sblk_aoff.end(), rout[t].aoff.begin(), rout[t].aoff.end());
			sblk_beg.insert(sblk_beg.end(), rout[t].beg.begin(), rout[t].beg.end());
		}
		sblk_beg.push_back(n_pairs);
	}
	const int64_t n_sblk = (int64_t)sblk_i1.size();
	h.n_sblk = n_sblk;

	// ---- sparse reduced system: the written blocks of S as an upper block-CSC structure (columns = i2,
	// rows i1 ascending, diagonal last): the block list above is row-major, a counting sort by column
	// keeps the rows ascending
	if(sparse_S) {
		Structure &ss = h.s_st;
		ss.nb = nc;
		ss.nnzb = n_sblk;
		ss.dim.assign(nc, dp);
		ss.base.resize(nc + 1);
		for(int64_t c = 0; c <= nc; ++ c)
			ss.base[c] = c * dp;
		ss.n = nc * dp;
		ss.col_ptr.assign(nc + 1, 0);
		for(int64_t b = 0; b < n_sblk; ++ b)
			++ ss.col_ptr[sblk_i2[b] + 1];
		for(int64_t c = 0; c < nc; ++ c)
			ss.col_ptr[c + 1] += ss.col_ptr[c];
		ss.row_idx.resize(n_sblk);
		ss.blk_off.resize(n_sblk);
		std::vector<int64_t> fill(ss.col_ptr.begin(), ss.col_ptr.end() - 1);
		h.sblk_voff.resize(n_sblk);
		for(int64_t b = 0; b < n_sblk; ++ b) {
			const int64_t q = fill[sblk_i2[b]] ++;
			ss.row_idx[q] = sblk_i1[b];
			ss.blk_off[q] = q * dp * dp;
			h.sblk_voff[b] = q * dp * dp;
		}
		ss.nvals = n_sblk * dp * dp;
		for(int64_t c = 0; c < nc; ++ c)
			SPP_REQUIRE(ss.col_ptr[c + 1] > ss.col_ptr[c] && ss.row_idx[ss.col_ptr[c + 1] - 1] == c, SPP_E_BADARG,
				"a pose without any diagonal contribution: the reduced system is singular");
	}

	clk.lap("block lists");
	// ---- work items: chunks of at most PAIR_CHUNK pairs
	std::vector<int32_t> item_blk, item_beg, item_end, item_slot;
	std::vector<int32_t> &multi_blk = h.multi_blk, &multi_ptr = h.multi_ptr;
	int32_t n_slots = 0;
	for(int64_t b = 0; b < n_sblk; ++ b) {
		const int64_t beg = sblk_beg[b], end = sblk_beg[b + 1];
		const int64_t nchunk = std::max<int64_t>(1, (end - beg + PAIR_CHUNK - 1) / PAIR_CHUNK);
		if(nchunk > 1) {
			multi_blk.push_back((int32_t)b);
			multi_ptr.push_back(n_slots);
		}
		for(int64_t c = 0; c < nchunk; ++ c) {
			item_blk.push_back((int32_t)b);
			item_beg.push_back((int32_t)(beg + c * PAIR_CHUNK));
			item_end.push_back((int32_t)std::min<int64_t>(end, beg + (c + 1) * PAIR_CHUNK));
			item_slot.push_back(nchunk > 1 ? n_slots ++ : -1);
		}
	}
	multi_ptr.push_back(n_slots);
	h.n_slots = n_slots;
	// Item order = execution order. Blocks are visited tile by tile (SACC_TILE x SACC_TILE cameras):
	// the W segments of the tile's row cameras and the U segments of its column cameras (~0.5 MB each
	// on Venice) then stay in the L2 of the XCD that works through the tile (the kernel hands each XCD
	// one contiguous range of items).
	bool interleave = false;
	std::vector<int32_t> xb_il;
	{
		static int64_t tb_env = -1;
		if(tb_env < 0) {
			const char *e = getenv("SPP_SACC_TILE"); // cameras per side of an item tile (1 = plain row-major block order)
			tb_env = e ? std::max<int64_t>(1, atol(e)) : 4;
		}
		static int64_t tbc_env = -1;
		if(tbc_env < 0) {
			const char *e = getenv("SPP_SACC_TILE_COLS"); // cameras per tile along a row of S (0: the whole row)
			tbc_env = e ? atol(e) : -2;
		}
		const int64_t TB = tb_env, TBC = (tbc_env == -2) ? TB : (tbc_env <= 0 ? nc : tbc_env), ntile = (nc + TBC - 1) / TBC;
		std::vector<int32_t> perm(item_blk.size());
		for(size_t q = 0; q < perm.size(); ++ q)
			perm[q] = (int32_t)q;
		static int il_env = -1;
		if(il_env < 0) {
			const char *e = getenv("SPP_SACC_XCD"); // 1: tiles dealt round-robin to the XCDs, 0: one contiguous range per XCD
			il_env = e ? atoi(e) : 1;
		}
		interleave = il_env != 0;
		std::vector<int64_t> tile_of_item(item_blk.size());
		for(size_t q = 0; q < tile_of_item.size(); ++ q)
			tile_of_item[q] = (sblk_i1[item_blk[q]] / TB) * ntile + sblk_i2[item_blk[q]] / TBC;
		auto tile_of = [&](int32_t q) { return tile_of_item[q]; };
		// (stable sorts by small keys: counting sorts -- two comparison sorts of the 180 000 items of the Venice shape were
		// 15 ms of the plan)
		auto stable_by_key = [&](std::vector<int32_t> &pm, int64_t n_keys, auto key_of) {
			if(n_keys > 8 * (int64_t)pm.size() + 1024) {
				std::stable_sort(pm.begin(), pm.end(), [&](int32_t x, int32_t y) { return key_of(x) < key_of(y); });
				return;
			}
			std::vector<int32_t> start((size_t)n_keys + 1, 0), out(pm.size());
			for(size_t q = 0; q < pm.size(); ++ q)
				++ start[key_of(pm[q]) + 1];
			for(int64_t k = 0; k < n_keys; ++ k)
				start[k + 1] += start[k];
			for(size_t q = 0; q < pm.size(); ++ q)
				out[start[key_of(pm[q])] ++] = pm[q];
			pm.swap(out);
		};
		stable_by_key(perm, ((nc + TB - 1) / TB) * ntile, tile_of);
		if(interleave) {
			// All eight XCDs work in the same neighbourhood of S: consecutive tiles (in tile-row-major order) go to
			// consecutive XCDs. A tile's camera segments still meet in ONE L2, and the blocks every XCD re-reads
			// within a band of S now share one working set in the memory-side cache (256 MB) instead of eight.
			// The deal is by WORK, not by count: the next tile goes to the XCD with the least work so far (a wave spends a
			// fixed cost per item plus one gather round per 64 pairs) -- on a banded S every eighth tile can be a diagonal
			// one, ten times as heavy as its neighbours (config 5 shape: 2.65 ms dealt by count, 1.5 ms by work).
			std::vector<int32_t> xcd_of(perm.size());
			int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for(size_t q = 0; q < perm.size();) {
				const int64_t t = tile_of(perm[q]);
				size_t e = q;
				int64_t cost = 0;
				for(; e < perm.size() && tile_of(perm[e]) == t; ++ e)
					cost += 2 + (item_end[perm[e]] - item_beg[perm[e]] + 63) / 64;
				int x = 0;
				for(int y = 1; y < 8; ++ y)
					if(load[y] < load[x])
						x = y;
				load[x] += cost;
				for(size_t i = q; i < e; ++ i)
					xcd_of[perm[i]] = x;
				q = e;
			}
			stable_by_key(perm, 8, [&](int32_t x) { return (int64_t)xcd_of[x]; });
			xb_il.assign(9, 0);
			for(size_t q = 0; q < perm.size(); ++ q)
				++ xb_il[xcd_of[q] + 1];
			for(int x = 0; x < 8; ++ x)
				xb_il[x + 1] += xb_il[x];
		}
		std::vector<int32_t> t_blk(perm.size()), t_beg(perm.size()), t_end(perm.size()), t_slot(perm.size());
		for(size_t q = 0; q < perm.size(); ++ q) {
			t_blk[q] = item_blk[perm[q]];
			t_beg[q] = item_beg[perm[q]];
			t_end[q] = item_end[perm[q]];
			t_slot[q] = item_slot[perm[q]];
		}
		item_blk.swap(t_blk); item_beg.swap(t_beg); item_end.swap(t_end); item_slot.swap(t_slot);
	}
	h.n_items = (int64_t)item_blk.size();
	// eight item ranges (one per XCD). Contiguous ranges: of equal WORK -- a wave spends a fixed cost per item plus
	// one gather round per 64 pairs; equal item counts would leave the last range ~45 % heavier
	{
		std::vector<int32_t> &xb = h.xb;
		xb.assign(9, 0);
		if(interleave)
			xb = xb_il;
		else {
			std::vector<int64_t> cost(h.n_items + 1, 0);
			for(int64_t q = 0; q < h.n_items; ++ q)
				cost[q + 1] = cost[q] + 2 + (item_end[q] - item_beg[q] + 63) / 64;
			for(int x = 1; x < 8; ++ x)
				xb[x] = (int32_t)(std::lower_bound(cost.begin(), cost.end(), cost[h.n_items] * x / 8) - cost.begin());
			xb[8] = (int32_t)h.n_items;
		}
		h.xcd_max_items = 0;
		for(int x = 0; x < 8; ++ x) {
			xb[x + 1] = std::max(xb[x + 1], xb[x]);
			h.xcd_max_items = std::max(h.xcd_max_items, xb[x + 1] - xb[x]);
		}
	}
	h.n_multi = (int64_t)multi_blk.size();

	// ---- rhs offsets
	h.pose_rbase.resize(nc);
	h.lm_rbase.resize(nl);
	for(int64_t c = 0; c < nc; ++ c)
		h.pose_rbase[c] = st.base[h.pose_block[c]];
	for(int64_t l = 0; l < nl; ++ l)
		h.lm_rbase[l] = st.base[h.lm_block[l]];

	clk.lap("work items");
	// self-contained item records (one 32-byte load per item in the kernel)
	h.recs.resize(item_blk.size());
	for(size_t q = 0; q < h.recs.size(); ++ q) {
		const int32_t b = item_blk[q];
		SaccItem &r = h.recs[q];
		r.beg = item_beg[q];
		r.end = item_end[q];
		r.pad = 0;
		r.aoff = -1;
		if(item_slot[q] >= 0) { // split block: s_multi_kernel sums the slots and adds A
			r.kind = 2;
			r.dst = (int64_t)item_slot[q] * dp * dp;
		} else {
			r.aoff = add_A ? sblk_aoff[b] : -1;
			if(sparse_S) {
				r.kind = 1;
				r.dst = h.sblk_voff[b];
			} else {
				r.kind = 0;
				r.dst = (int64_t)sblk_i1[b] * dp + (int64_t)sblk_i2[b] * dp * h.ld;
			}
		}
	}
	clk.lap("item records");
}

void build_schur_plan(spp_ctx *ctx, bool sparse_S, bool mis)
{
	const Structure &st = ctx->st;
	SchurPlan &sp = ctx->schur;
	sp.release_all();
	sp.sparse_S = sparse_S;
	hipStream_t s = ctx->stream;
	VClock clk0("schur plan (device)");
	std::unique_ptr<SchurPlanHost> hp(new SchurPlanHost);
	SchurPlanHost &h = *hp;
	clk0.lap("old plan released");
	schur_plan_host(st, ctx->shard_rank, ctx->shard_world, sparse_S, mis, h);
	VClock clk("schur plan (device)");
	const int dp = h.dp, dl = h.dl;
	const int64_t nc = h.nc, nl = h.nl, no = h.no;
	sp.dp = dp;
	sp.dl = dl;
	sp.pose_block.swap(h.pose_block);
	sp.lm_block.swap(h.lm_block);
	sp.is_lm.swap(h.is_lm);
	sp.nc = nc;
	sp.nl = nl;
	sp.nl_total = h.nl_total;
	sp.add_A = (ctx->shard_rank == 0);
	sp.n_red = h.n_red;
	sp.ld = h.ld;
	sp.no = no;
	sp.n_pairs = h.n_pairs;
	sp.n_sblk = h.n_sblk;
	sp.u_landmark_major = h.u_landmark_major;
	sp.factored = h.factored;
	sp.n_items = h.n_items;
	sp.n_multi = h.n_multi;
	sp.xcd_max_items = h.xcd_max_items;
	if(sparse_S) {
		sp.s_st = h.s_st;
		sp.sblk_voff.upload(h.sblk_voff, s);
	}

	// ---- upload
	sp.xcd_beg.upload(h.xb, s);
	sp.lm_ptr.upload(h.lm_ptr, s);
	sp.n_bs = h.bs_ptr.empty() ? 0 : (int64_t)h.bs_ptr.size() - 1;
	sp.bs_ptr.upload(h.bs_ptr, s);
	sp.lm_coff.upload(h.lm_coff, s);
	sp.lm_rbase.upload(h.lm_rbase, s);
	sp.obs_pose.upload(h.obs_pose, s);
	sp.obs_lm.upload(h.obs_lm, s);
	sp.obs_off.upload(h.obs_off, s);
	sp.pose_rbase.upload(h.pose_rbase, s);
	sp.cam_ptr.upload(h.cam_ptr, s);
	sp.cam_obs.upload(h.cam_obs, s);
	sp.items.upload(h.recs, s);
	sp.obs_wpos.upload(h.wpos, s);
	sp.sblk_i1.upload(h.sblk_i1, s);
	sp.sblk_i2.upload(h.sblk_i2, s);
	sp.sblk_aoff.upload(h.sblk_aoff, s);
	sp.pair_a.reserve(std::max<size_t>(1, h.pair_a.size()));
	sp.pair_b.reserve(std::max<size_t>(1, h.pair_b.size()));
	if(h.pair_a.size()) {
		SPP_HIP_CHECK(hipMemcpyAsync(sp.pair_a.p, h.pair_a.p, h.pair_a.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
		SPP_HIP_CHECK(hipMemcpyAsync(sp.pair_b.p, h.pair_b.p, h.pair_b.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
	}
	sp.multi_blk.upload(h.multi_blk, s);
	sp.multi_ptr.upload(h.multi_ptr, s);
	clk.lap("uploads enqueued");
	sp.cinv.reserve((size_t)std::max<int64_t>(1, nl) * dl * dl);
	sp.W.reserve((size_t)std::max<int64_t>(1, no) * dp * dl);
	if(sp.factored)
		sp.lfac.reserve((size_t)std::max<int64_t>(1, nl) * dl * dl);
	else
		sp.Up.reserve((size_t)std::max<int64_t>(1, no) * dp * dl);
	sp.xw.reserve((size_t)std::max<int64_t>(1, no) * dp);
	sp.partial.reserve((size_t)std::max<int32_t>(1, h.n_slots) * dp * dp);
	clk.lap("workspaces allocated");
	SPP_HIP_CHECK(hipStreamSynchronize(s)); // host vectors die here
	clk.lap("uploads done");

	if(!sparse_S)
		dense_reserve(ctx, (sp.n_red + DENSE_NB - 1) / DENSE_NB);
	clk.lap("dense workspaces");

	// ---- accounting (SURVEY 8d "Schur" + "Dense reduced solve")
	const double n = (double)sp.n_red;
	ctx->factor_flops = (int64_t)(n * n * n / 3.0 + 2.0 * n * n);
	ctx->factor_nnz = sp.ld * sp.ld;
	const int64_t blk_pl = 8 * dp * dl, blk_pp = 8 * dp * dp, blk_ll = 8 * dl * dl;
	ctx->solve_bytes = blk_pl * no + blk_ll * nl + blk_pp * h.n_ablk + 8 * st.n /* read */
		+ blk_pp * sp.n_sblk + 8 * st.n /* write S, solution */
		+ 8 * sp.n_red * sp.n_red /* dense factor touched once in place */;
	// the host image (pair lists, observation lists: ~250 MB on a Venice-sized problem, 8 ms to unmap) is released beside
	// the caller
	if(ctx->plan_trash.joinable())
		ctx->plan_trash.join();
	{
		SchurPlanHost *raw = hp.release();
		try {
			ctx->plan_trash = std::thread([raw]() { delete raw; });
		} catch(...) {
			delete raw;
		}
	}
	clk.lap("host plan handed to the releasing thread");
}

// host-only: the symbolic Schur plan of a structure, timed; out[0..7] = nc, nl, no, n_pairs, n_sblk, n_items, n_multi,
// a checksum of the pair lists and block list (tests compare thread counts against each other)
double schur_plan_host_probe(const Structure &st, int shard_rank, int shard_world, bool sparse_S, int64_t *out)
{
	const auto t0 = std::chrono::steady_clock::now();
	SchurPlanHost h;
	schur_plan_host(st, shard_rank, shard_world, sparse_S, false, h);
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	uint64_t sum = 1469598103934665603ull;
	auto mix = [&](uint64_t v) { sum = (sum ^ v) * 1099511628211ull; };
	for(size_t q = 0; q < h.pair_a.size(); ++ q)
		mix(((uint64_t)(uint32_t)h.pair_a[q] << 32) | (uint32_t)h.pair_b[q]);
	for(size_t q = 0; q < h.sblk_i1.size(); ++ q) {
		mix(((uint64_t)(uint32_t)h.sblk_i1[q] << 32) | (uint32_t)h.sblk_i2[q]);
		mix((uint64_t)h.sblk_aoff[q]);
	}
	for(size_t q = 0; q < h.recs.size(); ++ q) {
		mix(((uint64_t)(uint32_t)h.recs[q].beg << 32) | (uint32_t)h.recs[q].end);
		mix((uint64_t)h.recs[q].dst);
		mix((uint64_t)h.recs[q].aoff + (uint64_t)h.recs[q].kind);
	}
	for(size_t q = 0; q < h.xb.size(); ++ q)
		mix((uint64_t)h.xb[q]);
	out[0] = h.nc; out[1] = h.nl; out[2] = h.no; out[3] = h.n_pairs; out[4] = h.n_sblk; out[5] = h.n_items; out[6] = h.n_multi;
	out[7] = (int64_t)sum;
	return sec;
}

// --------------------------------------------------------------------------------------------------
// Approximate minimum degree on the block graph (A + A^T pattern of the upper-triangular input).
// Quotient graph with element absorption, approximate external degrees (|Le \ Lp| bound),
// mass elimination of indistinguishable variables is omitted (block graphs here are small).
// order[k] = block column eliminated k-th.
// --------------------------------------------------------------------------------------------------
void min_degree_order(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, std::vector<int64_t> &order)
{
	order.clear();
	order.reserve(nb);
	if(nb == 0)
		return;
	// adjacency (variables) of A + A^T without the diagonal
	std::vector<std::vector<int32_t> > adj(nb), elems(nb); // variable -> variable nbrs / element nbrs
	for(int64_t j = 0; j < nb; ++ j)
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
			int64_t i = row_idx[p];
			if(i != j) {
				adj[i].push_back((int32_t)j);
				adj[j].push_back((int32_t)i);
			}
		}
	for(int64_t i = 0; i < nb; ++ i) {
		std::sort(adj[i].begin(), adj[i].end());
		adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
	}
	std::vector<std::vector<int32_t> > elem_vars(nb); // element (named by its pivot) -> variables
	std::vector<uint8_t> state(nb, 0);               // 0 variable, 1 element, 2 absorbed element
	std::vector<int64_t> degree(nb);
	std::vector<int64_t> w(nb, -1);                  // |Le \ Lp| workspace, stamped
	std::vector<int64_t> mark(nb, -1);
	// degree buckets (doubly linked lists)
	std::vector<int32_t> head(nb + 1, -1), next(nb, -1), prev(nb, -1);
	auto bucket_insert = [&](int32_t v) {
		int64_t d = std::min<int64_t>(degree[v], nb);
		next[v] = head[d];
		prev[v] = -1;
		if(head[d] >= 0) prev[head[d]] = v;
		head[d] = v;
	};
	auto bucket_remove = [&](int32_t v) {
		int64_t d = std::min<int64_t>(degree[v], nb);
		if(prev[v] >= 0) next[prev[v]] = next[v]; else head[d] = next[v];
		if(next[v] >= 0) prev[next[v]] = prev[v];
	};
	for(int64_t i = 0; i < nb; ++ i) {
		degree[i] = (int64_t)adj[i].size();
		bucket_insert((int32_t)i);
	}
	int64_t mindeg = 0;
	std::vector<int32_t> Lp;
	for(int64_t step = 0; step < nb; ++ step) {
		while(mindeg <= nb && head[mindeg] < 0)
			++ mindeg;
		const int32_t p = head[mindeg];
		bucket_remove(p);
		order.push_back(p);
		// Lp = adj(p) U (union of Le for e in elems(p)) \ {p}
		Lp.clear();
		mark[p] = step;
		for(size_t q = 0; q < adj[p].size(); ++ q) {
			int32_t v = adj[p][q];
			if(state[v] == 0 && mark[v] != step) {
				mark[v] = step;
				Lp.push_back(v);
			}
		}
		for(size_t q = 0; q < elems[p].size(); ++ q) {
			int32_t e = elems[p][q];
			if(state[e] != 1)
				continue;
			for(size_t t = 0; t < elem_vars[e].size(); ++ t) {
				int32_t v = elem_vars[e][t];
				if(state[v] == 0 && mark[v] != step) {
					mark[v] = step;
					Lp.push_back(v);
				}
			}
			state[e] = 2; // absorbed into p
			std::vector<int32_t>().swap(elem_vars[e]);
		}
		state[p] = 1;
		std::vector<int32_t>().swap(adj[p]);
		std::vector<int32_t>().swap(elems[p]);
		elem_vars[p] = Lp;
		const int64_t lp = (int64_t)Lp.size();
		// w(e) = |Le \ Lp| for every element adjacent to a variable of Lp
		for(size_t q = 0; q < Lp.size(); ++ q) {
			int32_t v = Lp[q];
			for(size_t t = 0; t < elems[v].size(); ++ t) {
				int32_t e = elems[v][t];
				if(state[e] != 1)
					continue;
				if(w[e] < step * (nb + 1)) { // first touch in this step: stamp + live size
					int64_t sz = 0;
					for(size_t u = 0; u < elem_vars[e].size(); ++ u)
						if(state[elem_vars[e][u]] == 0)
							++ sz;
					w[e] = step * (nb + 1) + sz;
				}
				-- w[e];
			}
		}
		for(size_t q = 0; q < Lp.size(); ++ q) {
			const int32_t v = Lp[q];
			bucket_remove(v);
			// prune: variable neighbours inside Lp (now covered by element p) and dead entries
			{
				std::vector<int32_t> &a = adj[v];
				size_t o = 0;
				for(size_t t = 0; t < a.size(); ++ t)
					if(state[a[t]] == 0 && mark[a[t]] != step)
						a[o ++] = a[t];
				a.resize(o);
			}
			int64_t d = (int64_t)adj[v].size() + (lp - 1);
			{
				std::vector<int32_t> &el = elems[v];
				size_t o = 0;
				for(size_t t = 0; t < el.size(); ++ t) {
					int32_t e = el[t];
					if(state[e] != 1)
						continue;
					int64_t we = w[e] - step * (nb + 1);
					if(we <= 0) {
						// Le is a subset of Lp: aggressive absorption
						continue;
					}
					d += we;
					el[o ++] = e;
				}
				el.resize(o);
				el.push_back(p);
			}
			degree[v] = std::min<int64_t>(std::min<int64_t>(d, nb - step - 1), degree[v] + lp - 1);
			if(degree[v] < 0) degree[v] = 0;
			bucket_insert(v);
			if(degree[v] < mindeg)
				mindeg = degree[v];
		}
	}
}


// --------------------------------------------------------------------------------------------------
// Nested dissection on the block graph (George & Liu's automatic nested dissection): a level
// structure rooted at a pseudo-peripheral vertex, the smallest level of its middle part as vertex
// separator (thinned to the vertices that really touch the far side), recursion on the two parts,
// separator ordered last; subdomains of at most ND_LEAF blocks are ordered by minimum degree.
// Chain-like graphs (the reduced camera system of a long trajectory, BASELINE config 5) get an
// elimination tree of logarithmic height instead of the one long chain minimum degree produces, which
// is what the level-scheduled multifrontal kernels need (ordering).
// order[k] = block column eliminated k-th.
// --------------------------------------------------------------------------------------------------
static const int ND_LEAF = 48;

void nested_dissection_order(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, std::vector<int64_t> &order)
{
	order.assign(nb, -1);
	if(nb == 0)
		return;
	std::vector<std::vector<int32_t> > adj(nb);
	for(int64_t j = 0; j < nb; ++ j)
		for(int64_t p = col_ptr[j]; p < col_ptr[j + 1]; ++ p) {
			const int64_t i = row_idx[p];
			if(i != j) {
				adj[i].push_back((int32_t)j);
				adj[j].push_back((int32_t)i);
			}
		}
	// part[v]: id of the subset v currently belongs to; subsets are processed from a stack, each gets
	// a target range [lo, hi) of positions in the final order (separator at the end of the range)
	std::vector<int32_t> part(nb, 0), level(nb, -1), local(nb, -1);
	struct Task { std::vector<int32_t> verts; int64_t lo; };
	std::vector<Task> stack;
	{
		Task t;
		t.verts.resize(nb);
		for(int64_t v = 0; v < nb; ++ v)
			t.verts[v] = (int32_t)v;
		t.lo = 0;
		stack.push_back(std::move(t));
	}
	int32_t next_part = 1;
	std::vector<int32_t> queue;
	auto bfs = [&](int32_t root, int32_t pid, std::vector<int32_t> &out) { // level structure inside subset pid
		out.clear();
		out.push_back(root);
		level[root] = 0;
		for(size_t h = 0; h < out.size(); ++ h) {
			const int32_t v = out[h];
			for(size_t q = 0; q < adj[v].size(); ++ q) {
				const int32_t u = adj[v][q];
				if(part[u] == pid && level[u] < 0) {
					level[u] = level[v] + 1;
					out.push_back(u);
				}
			}
		}
	};
	while(!stack.empty()) {
		Task t = std::move(stack.back());
		stack.pop_back();
		const int64_t m = (int64_t)t.verts.size();
		if(m == 0)
			continue;
		const int32_t pid = part[t.verts[0]];
		if(m <= ND_LEAF) {
			// minimum degree on the induced subgraph (upper pattern in local numbering)
			for(int64_t q = 0; q < m; ++ q)
				local[t.verts[q]] = (int32_t)q;
			std::vector<int64_t> cp(m + 1, 0), ri;
			for(int64_t q = 0; q < m; ++ q) {
				const int32_t v = t.verts[q];
				for(size_t e = 0; e < adj[v].size(); ++ e) {
					const int32_t u = adj[v][e];
					if(part[u] == pid && local[u] < q)
						ri.push_back(local[u]);
				}
				std::sort(ri.begin() + cp[q], ri.end());
				ri.erase(std::unique(ri.begin() + cp[q], ri.end()), ri.end());
				ri.push_back(q); // diagonal last
				cp[q + 1] = (int64_t)ri.size();
			}
			std::vector<int64_t> lo;
			min_degree_order(m, cp.data(), ri.data(), lo);
			for(int64_t q = 0; q < m; ++ q)
				order[t.lo + q] = t.verts[lo[q]];
			for(int64_t q = 0; q < m; ++ q) {
				local[t.verts[q]] = -1;
				part[t.verts[q]] = -1; // done
			}
			continue;
		}
		// pseudo-peripheral root: two sweeps
		bfs(t.verts[0], pid, queue);
		if((int64_t)queue.size() < m) {
			// disconnected: split off this component, no separator
			const int32_t pa = next_part ++;
			Task a, b;
			for(size_t q = 0; q < queue.size(); ++ q)
				part[queue[q]] = pa;
			for(int64_t q = 0; q < m; ++ q) {
				const int32_t v = t.verts[q];
				level[v] = -1;
				(part[v] == pa ? a : b).verts.push_back(v);
			}
			a.lo = t.lo;
			b.lo = t.lo + (int64_t)a.verts.size();
			stack.push_back(std::move(a));
			stack.push_back(std::move(b));
			continue;
		}
		int32_t far = queue.back();
		for(size_t q = 0; q < queue.size(); ++ q)
			level[queue[q]] = -1;
		bfs(far, pid, queue);
		const int32_t nlev = level[queue.back()] + 1;
		if(nlev < 3) { // clique-like: no useful separator
			for(int64_t q = 0; q < m; ++ q) {
				local[t.verts[q]] = (int32_t)q;
				level[t.verts[q]] = -1;
			}
			std::vector<int64_t> cp(m + 1, 0), ri;
			for(int64_t q = 0; q < m; ++ q) {
				const int32_t v = t.verts[q];
				for(size_t e = 0; e < adj[v].size(); ++ e) {
					const int32_t u = adj[v][e];
					if(part[u] == pid && local[u] < q)
						ri.push_back(local[u]);
				}
				std::sort(ri.begin() + cp[q], ri.end());
				ri.erase(std::unique(ri.begin() + cp[q], ri.end()), ri.end());
				ri.push_back(q);
				cp[q + 1] = (int64_t)ri.size();
			}
			std::vector<int64_t> lo;
			min_degree_order(m, cp.data(), ri.data(), lo);
			for(int64_t q = 0; q < m; ++ q)
				order[t.lo + q] = t.verts[lo[q]];
			for(int64_t q = 0; q < m; ++ q) {
				local[t.verts[q]] = -1;
				part[t.verts[q]] = -1;
			}
			continue;
		}
		// level sizes; separator = smallest level whose cumulative position lies in the middle part
		std::vector<int64_t> lsize(nlev, 0);
		for(size_t q = 0; q < queue.size(); ++ q)
			++ lsize[level[queue[q]]];
		int32_t best = -1;
		{
			int64_t below = 0;
			double best_cost = 0;
			for(int32_t l = 0; l < nlev; ++ l) {
				const int64_t above = m - below - lsize[l];
				if(l > 0 && l + 1 < nlev) {
					const double bal = (double)std::min(below, above) / (double)std::max<int64_t>(1, std::max(below, above));
					// small separators, balanced parts: separator size penalized by imbalance
					const double cost = (double)lsize[l] * (1.0 + 2.0 * (1.0 - bal) * (1.0 - bal) * 4.0);
					if(bal >= 0.25 && (best < 0 || cost < best_cost)) {
						best = l;
						best_cost = cost;
					}
				}
				below += lsize[l];
			}
			if(best < 0)
				best = nlev / 2;
		}
		// near part: levels < best; far part: levels > best; separator vertices without a neighbour in
		// level best + 1 move to the near part
		const int32_t pa = next_part ++, pb = next_part ++;
		Task a, b;
		std::vector<int32_t> sep;
		for(size_t q = 0; q < queue.size(); ++ q) {
			const int32_t v = queue[q];
			if(level[v] < best)
				a.verts.push_back(v);
			else if(level[v] > best)
				b.verts.push_back(v);
			else {
				bool touches = false;
				for(size_t e = 0; e < adj[v].size() && !touches; ++ e) {
					const int32_t u = adj[v][e];
					touches = part[u] == pid && level[u] == best + 1;
				}
				if(touches)
					sep.push_back(v);
				else
					a.verts.push_back(v);
			}
		}
		for(size_t q = 0; q < a.verts.size(); ++ q)
			part[a.verts[q]] = pa;
		for(size_t q = 0; q < b.verts.size(); ++ q)
			part[b.verts[q]] = pb;
		for(size_t q = 0; q < queue.size(); ++ q)
			level[queue[q]] = -1;
		a.lo = t.lo;
		b.lo = t.lo + (int64_t)a.verts.size();
		const int64_t slo = b.lo + (int64_t)b.verts.size();
		for(size_t q = 0; q < sep.size(); ++ q) {
			order[slo + (int64_t)q] = sep[q];
			part[sep[q]] = -1;
		}
		stack.push_back(std::move(a));
		stack.push_back(std::move(b));
	}
}

} // namespace spp
