// spp_sparse.hip -- supernodal multifrontal block Cholesky (R^T R = P Lambda P^T, R upper) and the two
// triangular solves on gfx950, fp64: the sparse path for pose graphs (configs 1, 2) and the fallback
// whenever guided Schur does not apply.
//
// Replaces CLinearSolver_UberBlock::Solve_PosDef_Blocky (reference include/slam/LinearSolver_UberBlock.h:312-426):
//   p_BlockOrdering (AMD)                   src/slam/OrderingMagic.cpp:701       -> min_degree_order (ours, host)
//   Permute_UpperTriangular_To              src/slam/BlockMatrix.cpp:8183        -> index lists, no data moved
//   Build_EliminationTree / n_Build_EReach  src/slam/BlockMatrix.cpp:9403,9453   -> etree + postorder + supernodes (host, once)
//   CholeskyOf_FBS (up-looking, serial)     src/slam/BlockMatrixFBS.inl:2342     -> multifrontal fronts, level-parallel
//   UpperTriangularTranspose_Solve_FBS      src/slam/BlockMatrixFBS.inl:2136     -> inside the factorization (slot column)
//   UpperTriangular_Solve_FBS               src/slam/BlockMatrixFBS.inl:2233     -> bwd_kernel (root -> leaves)
//   (Inverse)Permute_LeftHandSide_Vector    src/slam/BlockMatrix.cpp:9323,9379   -> gather/scatter by scalar permutation
//
// The reference factors column by column with sorted-list merges (no supernodes, no BLAS3, serial:
// SURVEY 8a-7). Here every supernode owns a dense frontal matrix F (h x h, upper part used, column
// major) resident in HBM for the whole solve (288 GB: no stack juggling):
//   F <- blocks of Lambda in the supernode's rows  (+) extend-add of the children's update matrices
//   F11 = R11^T R11 ; R12 = R11^-T F12 ; F22 -= R12^T R12         (partial dense factorization)
// All fronts of one level of the assembly tree are processed by ONE launch (a workgroup per front);
// fronts too large for one workgroup go through the multi-workgroup dense kernels of spp_dense.hip
// (MFMA trailing update). Children are summed into their parent in a fixed order and nothing uses
// atomics, so the factor is bit-reproducible.
//
// Roofline: small fronts are latency/HBM bound (SURVEY 7.3 "latency, not bandwidth, for pose
// graphs"); only the large fronts near the root reach the MFMA path.

#include "spp_internal.h"
#include "spp_tiles.h"
#include "spp_dense_dev.h"
#include <algorithm>
#include <thread>
#include <exception>
#include <numeric>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace spp {

struct SparsePlan {
	DevBuf<unsigned char> index_store;         // the one allocation behind the index arrays below (UploadArena)
	int64_t nb = 0, n = 0;
	int64_t n_snodes = 0, n_levels = 0;
	int64_t front_doubles = 0, vbuf_doubles = 0;
	// host
	std::vector<int32_t> h_level_ptr;          // [n_levels+1]
	std::vector<int32_t> h_front_h, h_front_w; // per supernode (scalar sizes)
	std::vector<int64_t> h_front_off;
	std::vector<int32_t> h_level_fronts;
	std::vector<int32_t> h_front_ld, h_front_pad, h_front_cls;
	std::vector<int32_t> h_cls_ptr;            // [n_levels * NCLS + 1]: fronts of (level, class) in level_fronts
	std::vector<int32_t> h_child_ptr, h_child_list, h_asm_ptr;
	// device
	DevBuf<int32_t> level_fronts;              // fronts grouped by level
	DevBuf<int64_t> front_off;                 // [ns] offset of F in `fronts`
	DevBuf<int32_t> front_h, front_w, front_ld; // [ns]
	DevBuf<int32_t> front_pad;                 // [ns] identity padding inserted after the pivot block (big fronts)
	DevBuf<int64_t> front_voff;                // [ns] offset of the solve work vector
	DevBuf<int32_t> asm_ptr;                   // [ns+1]
	DevBuf<int64_t> asm_src;                   // [n_asm] (offset in vals << 1) | transpose
	DevBuf<int32_t> asm_dst;                   // [n_asm] local scalar row | local scalar col << 16
	DevBuf<int32_t> asm_shape;                 // [n_asm] dst rows | dst cols << 8
	DevBuf<int32_t> child_ptr;                 // [ns+1]
	DevBuf<int32_t> child_list;                // children in fixed (ascending) order
	DevBuf<int32_t> rel_ptr;                   // [ns+1] into rel (per supernode as a CHILD)
	DevBuf<int32_t> rel;                       // parent-local scalar index of each update row of the child
	DevBuf<int32_t> rows_ptr;                  // [ns+1]
	DevBuf<int32_t> rows;                      // permuted global scalar index of each local row
	DevBuf<int32_t> perm_scalar;               // [n] permuted scalar -> original scalar
	DevBuf<double> fronts, vbuf, xperm;
	DevBuf<int> info;
	// dependency-driven launches (front_dag_kernel): the fronts of levels < dag_level_limit, children first
	DevBuf<int32_t> front_cls, front_parent, front_level; // [ns]
	DevBuf<int32_t> dag_list, dag_list_bwd;    // dispatch order of the factorization / of the backward substitution
	DevBuf<int> dag_done;                      // [ns] epoch flags
	int32_t dag_n = 0, dag_n_bwd = 0, dag_level_limit = 0;
	int dag_solves = 0;                        // factorizations run on the team barrier counters so far
	DevBuf<int32_t> dag_rank, front_team;      // per block of the factorization launch / per front
	DevBuf<int> team_bar;                      // [ns] barrier counters of the teams (monotonic)
	DevBuf<int64_t> front_tinv;                // [ns] offset of a big front's diagonal-block inverses
	DevBuf<double> team_tinv;
	int32_t dag_first1 = 0;                    // position in dag_list of the first front of level 1
	// the levels below the first front of more than 64 rows go as a launch of their own with 256 threads and the LDS of
	// their own classes: four workgroups per CU instead of one (the wide bottom of a large tree)
	int32_t dag_split = 0, dag_split_level = 0;
	size_t dag_lds_low = 0;
	int dag_epoch = 0;
	size_t dag_lds = 0;                        // dynamic LDS of the factorization launch (largest class present)
	bool dag_ok = true;                        // false after a timed-out wait: level-by-level launches from then on
};

// --------------------------------------------------------------------------------------------------
// host symbolic
// --------------------------------------------------------------------------------------------------
namespace {

struct Sym {
	int64_t nb;
	std::vector<int64_t> order, inv;            // final elimination order and its inverse
	std::vector<std::vector<int32_t> > up;      // permuted upper pattern: rows (< col) of each column
	std::vector<int32_t> parent;
};

static void etree_of(int64_t nb, const std::vector<std::vector<int32_t> > &up, std::vector<int32_t> &parent)
{
	std::vector<int32_t> anc(nb, -1);
	parent.assign(nb, -1);
	for(int64_t j = 0; j < nb; ++ j)
		for(size_t q = 0; q < up[j].size(); ++ q) {
			int32_t i = up[j][q];
			while(i != -1 && i < j) {
				int32_t next = anc[i];
				anc[i] = (int32_t)j;
				if(next == -1)
					parent[i] = (int32_t)j;
				i = next;
			}
		}
}

static void permuted_pattern(const Structure &st, const std::vector<int64_t> &inv, std::vector<std::vector<int32_t> > &up)
{
	up.assign(st.nb, std::vector<int32_t>());
	for(int64_t j = 0; j < st.nb; ++ j)
		for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
			int64_t i = st.row_idx[p];
			if(i == j)
				continue;
			int64_t a = inv[i], b = inv[j];
			if(a > b) std::swap(a, b);
			up[b].push_back((int32_t)a);
		}
	for(int64_t j = 0; j < st.nb; ++ j)
		std::sort(up[j].begin(), up[j].end());
}

static void postorder(const std::vector<int32_t> &parent, std::vector<int32_t> &post)
{
	const int64_t nb = (int64_t)parent.size();
	std::vector<int32_t> head(nb, -1), next(nb, -1);
	for(int64_t j = nb; j > 0;) { // children lists in ascending order
		-- j;
		if(parent[j] >= 0) {
			next[j] = head[parent[j]];
			head[parent[j]] = (int32_t)j;
		}
	}
	post.clear();
	post.reserve(nb);
	std::vector<int32_t> stack;
	for(int64_t r = 0; r < nb; ++ r) {
		if(parent[r] >= 0)
			continue;
		stack.push_back((int32_t)r);
		while(!stack.empty()) {
			int32_t v = stack.back();
			int32_t c = head[v];
			if(c >= 0) {
				head[v] = next[c];
				stack.push_back(c);
			} else {
				stack.pop_back();
				post.push_back(v);
			}
		}
	}
}

} // namespace

void sparse_release(spp_ctx *ctx)
{
	delete ctx->sparse;
	ctx->sparse = nullptr;
}

void sparse_dag_disable(spp_ctx *ctx)
{
	if(ctx->sparse)
		ctx->sparse->dag_ok = false;
}

int64_t sparse_info(const spp_ctx *ctx, int what)
{
	if(!ctx->sparse)
		return 0;
	if(what == SPP_INFO_N_SUPERNODES) return ctx->sparse->n_snodes;
	if(what == SPP_INFO_N_LEVELS) return ctx->sparse->n_levels;
	return 0;
}

static const int NCLS = 5;
// big fronts inside the dependency-driven launch (bigfront_team_body): threads and LDS of a team member
constexpr int TEAM_THREADS = 1024;
constexpr size_t TEAM_LDS_DOUBLES = (size_t)(NB + 16) * FS_STRIDE > (size_t)POTRF_LDS_DOUBLES ? (size_t)(NB + 16) * FS_STRIDE : (size_t)POTRF_LDS_DOUBLES;
static const int MID_FRONT_MAX = 640; // largest padded height the one-workgroup in-place (HBM image) kernel is built for
static const int MID_FRONT_DEFAULT = 320; // default split: fronts above go to the dense MFMA kernels

void sparse_analyze(spp_ctx *ctx, const Structure &st)
{
	const int64_t nb = st.nb;
	// fronts above this padded height go to the dense MFMA kernels (one after the other, many workgroups
	// each) instead of the one-workgroup in-place kernel; SPP_MID_FRONT_MAX tunes the split (128 .. 640; with the single-stream dense steps 320 measured best on sphere2500, neutral on the other sparse workloads)
	int mid_front_max = MID_FRONT_DEFAULT;
	if(const char *e = getenv("SPP_MID_FRONT_MAX"))
		mid_front_max = std::max(128, std::min(MID_FRONT_MAX, atoi(e)));
	sparse_release(ctx);
	SparsePlan *sp = new SparsePlan;
	ctx->sparse = sp;
	sp->nb = nb;
	sp->n = st.n;
	SPP_REQUIRE(st.n < (int64_t(1) << 31), SPP_E_UNSUPPORTED, "sparse path: scalar dimension exceeds 32-bit indices");

	// ---- 1. + 2. fill-reducing order, postorder of its elimination tree, row structure of every row
	// of R (= column structure of L, children merged upward): rstruct[j] = sorted block columns c > j
	// with R(j, c) != 0
	VClock clk("sparse_analyze");
	struct Symbolic {
		std::vector<int64_t> order;
		std::vector<std::vector<int32_t> > rstruct;
		std::vector<int32_t> parent;
		double flops = 0;   // sum_j (row count)^2, block level
		int64_t height = 0; // longest root-to-leaf path of the elimination tree, in block columns
	};
	auto symbolic = [&st, nb](const std::vector<int64_t> &order0, Symbolic &y) {
		std::vector<int64_t> inv(nb);
		std::vector<std::vector<int32_t> > up;
		std::vector<int32_t> post;
		for(int64_t k = 0; k < nb; ++ k)
			inv[order0[k]] = k;
		permuted_pattern(st, inv, up);
		etree_of(nb, up, y.parent);
		postorder(y.parent, post);
		y.order.resize(nb);
		for(int64_t k = 0; k < nb; ++ k)
			y.order[k] = order0[post[k]];
		for(int64_t k = 0; k < nb; ++ k)
			inv[y.order[k]] = k;
		permuted_pattern(st, inv, up);
		etree_of(nb, up, y.parent);
		std::vector<std::vector<int32_t> > &rstruct = y.rstruct;
		rstruct.assign(nb, std::vector<int32_t>());
		// entries of A in row j right of the diagonal: from the permuted upper pattern, (a, b) a < b
		for(int64_t b = 0; b < nb; ++ b)
			for(size_t q = 0; q < up[b].size(); ++ q)
				rstruct[up[b][q]].push_back((int32_t)b);
		y.flops = 0;
		std::vector<int32_t> depth(nb, 1);
		for(int64_t j = 0; j < nb; ++ j) {
			std::vector<int32_t> &rs = rstruct[j];
			std::sort(rs.begin(), rs.end());
			rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
			y.flops += (double)(rs.size() + 1) * (double)(rs.size() + 1);
			// push to the parent: struct[parent] U= struct[j] \ {parent}
			const int32_t p = y.parent[j];
			SPP_REQUIRE(p < 0 || (!rs.empty() && rs[0] == p), SPP_E_HIP, "internal: etree/structure mismatch");
			if(p >= 0) {
				// ps is not sorted yet (raw A entries): append, it is sorted/uniqued when p is visited
				std::vector<int32_t> &ps = rstruct[p];
				ps.insert(ps.end(), rs.begin() + 1, rs.end());
				depth[p] = std::max(depth[p], depth[j] + 1);
			}
		}
		y.height = 0;
		for(int64_t j = 0; j < nb; ++ j)
			y.height = std::max<int64_t>(y.height, depth[j]);
	};
	Symbolic chosen_sym;
	{
		// Minimum degree gives the least fill; on chain-like graphs its elimination tree is one long
		// chain, though, and the level-scheduled kernels then run one small front after the other.
		// Nested dissection is taken instead when it cuts the tree height by more than 4x at no more
		// than 3x the block-level flops, or when it is simply better on both counts.
		// SPP_ORDERING = amd | nd forces one of them. The two candidates (ordering + symbolic factorization each) are
		// independent: they run side by side on two host threads.
		const char *force = getenv("SPP_ORDERING");
		const bool want_nd = force ? !strcmp(force, "nd") : (nb >= 256);
		Symbolic y_amd, y_nd;
		std::exception_ptr err_nd;
		std::thread t_nd;
		if(want_nd)
			t_nd = std::thread([&]() {
				try {
					std::vector<int64_t> nd;
					nested_dissection_order(nb, st.col_ptr.data(), st.row_idx.data(), nd);
					symbolic(nd, y_nd);
				} catch(...) {
					err_nd = std::current_exception();
				}
			});
		try {
			std::vector<int64_t> amd;
			min_degree_order(nb, st.col_ptr.data(), st.row_idx.data(), amd);
			symbolic(amd, y_amd);
		} catch(...) {
			if(t_nd.joinable())
				t_nd.join();
			throw;
		}
		if(t_nd.joinable())
			t_nd.join();
		if(err_nd)
			std::rethrow_exception(err_nd);
		bool take = false;
		if(want_nd) {
			take = force ? true : ((4 * y_nd.height < y_amd.height && y_nd.flops <= 3.0 * y_amd.flops) ||
				(y_nd.height < y_amd.height && y_nd.flops <= 1.02 * y_amd.flops));
			if(getenv("SPP_VERBOSE"))
				fprintf(stderr, "[spp] ordering: amd height %lld flops %.3g | nd height %lld flops %.3g -> %s\n",
					(long long)y_amd.height, y_amd.flops, (long long)y_nd.height, y_nd.flops, take ? "nd" : "amd");
		}
		chosen_sym = std::move(take ? y_nd : y_amd);
	}
	std::vector<int64_t> &order = chosen_sym.order;
	std::vector<std::vector<int32_t> > &rstruct = chosen_sym.rstruct;
	std::vector<int32_t> &parent = chosen_sym.parent;
	const double sym_flops = chosen_sym.flops;
	(void)sym_flops;
	const int64_t sym_height = chosen_sym.height;
	(void)sym_height;
	std::vector<int64_t> inv(nb);
	for(int64_t k = 0; k < nb; ++ k)
		inv[order[k]] = k;
	clk.lap("orderings + symbolic factorization");
	ctx->order = order;

	// ---- 3. fundamental supernodes + relaxed amalgamation of the last child
	std::vector<int32_t> n_child(nb, 0);
	for(int64_t j = 0; j < nb; ++ j)
		if(parent[j] >= 0)
			++ n_child[parent[j]];
	std::vector<int32_t> sn_first; // first block column of each supernode
	std::vector<int32_t> sn_of(nb);
	for(int64_t j = 0; j < nb; ++ j) {
		bool join = j > 0 && parent[j - 1] == j && n_child[j] == 1 &&
			rstruct[j - 1].size() == rstruct[j].size() + 1;
		if(!join)
			sn_first.push_back((int32_t)j);
		sn_of[j] = (int32_t)sn_first.size() - 1;
	}
	// relaxed amalgamation: merge supernode s into the supernode that starts right after it when
	// that one is its parent and the explicit zeros stay small
	{
		std::vector<int32_t> pdim(nb);
		for(int64_t k = 0; k < nb; ++ k)
			pdim[k] = st.dim[order[k]];
		std::vector<int32_t> first2;
		int64_t ns = (int64_t)sn_first.size();
		std::vector<int32_t> sn_last(ns);
		for(int64_t s = 0; s < ns; ++ s)
			sn_last[s] = (s + 1 < ns ? sn_first[s + 1] : (int32_t)nb) - 1;
		// scalar width / height helpers on current (possibly merged) supernodes
		std::vector<char> merged_into_next(ns, 0);
		// process from the leaves: greedy chain merging
		std::vector<int64_t> cur_w(ns), cur_hbeyond(ns), cur_zeros(ns, 0); // cur_zeros: explicit zeros already bought
		for(int64_t s = 0; s < ns; ++ s) {
			int64_t w = 0;
			for(int32_t c = sn_first[s]; c <= sn_last[s]; ++ c)
				w += pdim[c];
			int64_t hb = 0;
			const std::vector<int32_t> &rs = rstruct[sn_last[s]];
			for(size_t q = 0; q < rs.size(); ++ q)
				hb += pdim[rs[q]];
			cur_w[s] = w;
			cur_hbeyond[s] = hb;
		}
		for(int64_t s = 0; s + 1 < ns; ++ s) {
			const int32_t last = sn_last[s];
			if(parent[last] != last + 1)
				continue; // the next supernode is not the parent
			const int64_t wp = cur_w[s + 1], hp = wp + cur_hbeyond[s + 1];
			const int64_t ws = cur_w[s], hs_beyond = cur_hbeyond[s];
			// rows of s beyond its pivot block are a subset of the parent's rows (incl. its pivot block)
			const int64_t zeros = ws * (hp - hs_beyond);
			const int64_t merged_panel = (ws + wp) * (ws + hp);
			static int64_t small_w = -1;
			static double zero_frac = 0.12;
			if(small_w < 0) {
				const char *e = getenv("SPP_AMALG_SMALL"); // merged pivot widths up to this are always accepted
				small_w = e ? atol(e) : 32;
				if(const char *z = getenv("SPP_AMALG_ZEROS"))
					zero_frac = atof(z);
			}
			const bool small = (ws + wp) <= small_w;
			// ALL explicit zeros of the merged supernode count (those bought by earlier merges included):
			// along a chain -- the elimination tree of a banded system -- the newly added zeros alone
			// always look small next to the growing panel, and the whole chain would collapse into one
			// dense front
			const int64_t ztot = zeros + cur_zeros[s] + cur_zeros[s + 1];
			// (the looser bound only while the merged front stays small: a front of up to ~100 rows costs its latency, ~13 us,
			// whatever its size, so fewer levels pay; above that the explicit zeros cost flops on the critical path)
			static int64_t relax_h = -1;
			static double zero_frac_small = 0.2;
			if(relax_h < 0) {
				const char *e = getenv("SPP_AMALG_RELAX_H");
				relax_h = e ? atol(e) : 96;
				if(const char *z = getenv("SPP_AMALG_ZEROS_SMALL"))
					zero_frac_small = atof(z);
			}
			const double zf = (ws + hp <= relax_h) ? std::max(zero_frac, zero_frac_small) : zero_frac;
			if(zeros == 0 || small || (double)ztot <= zf * (double)merged_panel) {
				merged_into_next[s] = 1;
				cur_w[s + 1] = ws + wp; // the merged supernode takes the parent's slot
				cur_zeros[s + 1] = ztot;
			}
		}
		for(int64_t s = 0; s < ns; ++ s)
			if(s == 0 || !merged_into_next[s - 1])
				first2.push_back(sn_first[s]);
		sn_first.swap(first2);
		for(size_t s = 0; s < sn_first.size(); ++ s) {
			int32_t e = (s + 1 < sn_first.size()) ? sn_first[s + 1] : (int32_t)nb;
			for(int32_t c = sn_first[s]; c < e; ++ c)
				sn_of[c] = (int32_t)s;
		}
	}
	const int64_t ns = (int64_t)sn_first.size();
	sp->n_snodes = ns;

	clk.lap("supernodes + amalgamation");
	// ---- 4. per supernode: block row structure (own columns, then the union of the rows beyond)
	std::vector<int32_t> pdim(nb);
	std::vector<int64_t> pbase(nb + 1, 0);
	for(int64_t k = 0; k < nb; ++ k) {
		pdim[k] = st.dim[order[k]];
		pbase[k + 1] = pbase[k] + pdim[k];
	}
	std::vector<std::vector<int32_t> > sn_rows(ns); // block columns of the front, ascending
	std::vector<int32_t> sn_parent(ns, -1), sn_ncols(ns);
	for(int64_t s = 0; s < ns; ++ s) {
		const int32_t c0 = sn_first[s], c1 = (s + 1 < ns) ? sn_first[s + 1] : (int32_t)nb;
		sn_ncols[s] = c1 - c0;
		std::vector<int32_t> &r = sn_rows[s];
		for(int32_t c = c0; c < c1; ++ c)
			r.push_back(c);
		std::vector<int32_t> beyond;
		for(int32_t c = c0; c < c1; ++ c)
			for(size_t q = 0; q < rstruct[c].size(); ++ q)
				if(rstruct[c][q] >= c1)
					beyond.push_back(rstruct[c][q]);
		std::sort(beyond.begin(), beyond.end());
		beyond.erase(std::unique(beyond.begin(), beyond.end()), beyond.end());
		r.insert(r.end(), beyond.begin(), beyond.end());
		if(!beyond.empty())
			sn_parent[s] = sn_of[beyond[0]];
	}

	// ---- 5. levels of the assembly tree
	std::vector<int32_t> level(ns, 0);
	int32_t max_level = 0;
	for(int64_t s = 0; s < ns; ++ s) { // children precede parents (postorder)
		if(sn_parent[s] >= 0) {
			SPP_REQUIRE(sn_parent[s] > s, SPP_E_HIP, "internal: supernode order");
			level[sn_parent[s]] = std::max(level[sn_parent[s]], level[s] + 1);
		}
		max_level = std::max(max_level, level[s]);
	}
	sp->n_levels = max_level + 1;
	sp->h_level_ptr.assign(sp->n_levels + 1, 0);
	for(int64_t s = 0; s < ns; ++ s)
		++ sp->h_level_ptr[level[s] + 1];
	for(int64_t l = 0; l < sp->n_levels; ++ l)
		sp->h_level_ptr[l + 1] += sp->h_level_ptr[l];
	sp->h_level_fronts.resize(ns);
	{
		std::vector<int32_t> fill(sp->h_level_ptr.begin(), sp->h_level_ptr.end() - 1);
		for(int64_t s = 0; s < ns; ++ s)
			sp->h_level_fronts[fill[level[s]] ++] = (int32_t)s;
	}

	clk.lap("front structures + levels");
	// ---- 6. flat arrays
	std::vector<int32_t> front_h(ns), front_w(ns), front_ld(ns), front_pad(ns), front_cls(ns), rows_ptr(ns + 1, 0), rows;
	std::vector<int64_t> front_off(ns), front_voff(ns);
	std::vector<std::vector<int32_t> > loc_off(ns); // local scalar offset of each block row of the front
	int64_t foff = 0, voff = 0;
	double flops = 0;
	int64_t nnz_r = 0;
	for(int64_t s = 0; s < ns; ++ s) {
		int32_t h = 0, w = 0;
		loc_off[s].resize(sn_rows[s].size() + 1);
		for(size_t q = 0; q < sn_rows[s].size(); ++ q) {
			loc_off[s][q] = h;
			h += pdim[sn_rows[s][q]];
			if((int32_t)q < sn_ncols[s])
				w = h;
		}
		loc_off[s][sn_rows[s].size()] = h;
		SPP_REQUIRE(h < 32767, SPP_E_UNSUPPORTED, "front too large for 16-bit local indices");
		// The right-hand side rides along as ONE EXTRA non-pivot row / column of every front (local index h_real):
		// its column holds b on the pivot rows, the factorization's row-panel and update steps turn that into y and
		// into the residual handed to the parent (the extend-add maps the child's slot onto the parent's), so the
		// forward substitution R^T y = b costs no launch of its own -- as in the dense factor (padding column n).
		const int32_t h_real = h;
		h = h_real + 1;
		front_h[s] = h;
		front_w[s] = w;
		// size class: the in-LDS kernels pad the pivot block to a multiple of 16 inside their image
		// (hp16 <= 32 / 64 / 128); larger fronts live padded in HBM: pivot block rounded up to 128
		// with identity so that the dense 128-block kernels apply unchanged
		const int32_t w16 = (w + 15) & ~15, hp16 = (w16 + (h - w) + 15) & ~15;
		int32_t cls = hp16 <= 32 ? 0 : (hp16 <= 64 ? 1 : (hp16 <= 128 ? 2 : (hp16 <= mid_front_max ? 3 : 4)));
		front_cls[s] = cls;
		front_pad[s] = (cls == 4) ? (((w + 127) & ~127) - w) : (cls == 3 ? w16 - w : 0);
		// class 3 works on whole 16 x 16 tiles in place: its HBM image is rounded up to tiles
		const int32_t hp = (cls == 3) ? hp16 : h + front_pad[s];
		front_ld[s] = (hp + 1) & ~1; // even: 16-byte aligned columns
		front_off[s] = foff;
		foff += (int64_t)front_ld[s] * hp;
		foff = (foff + 1) & ~int64_t(1);
		front_voff[s] = voff;
		voff += h;
		rows_ptr[s + 1] = rows_ptr[s] + h;
		for(size_t q = 0; q < sn_rows[s].size(); ++ q)
			for(int32_t e = 0; e < pdim[sn_rows[s][q]]; ++ e)
				rows.push_back((int32_t)(pbase[sn_rows[s][q]] + e));
		rows.push_back(0); // the right-hand-side slot has no global row (never dereferenced)
		// flops of the partial factorization: sum over pivots of (remaining width)^2 (the matrix alone)
		for(int32_t j = 0; j < w; ++ j)
			flops += (double)(h_real - j) * (double)(h_real - j);
		nnz_r += (int64_t)w * h_real - (int64_t)w * (w - 1) / 2;
	}
	// regroup the level lists by size class
	sp->h_cls_ptr.assign(sp->n_levels * NCLS + 1, 0);
	for(int64_t q = 0; q < ns; ++ q)
		++ sp->h_cls_ptr[level[q] * NCLS + front_cls[q] + 1];
	for(int64_t q = 0; q < sp->n_levels * NCLS; ++ q)
		sp->h_cls_ptr[q + 1] += sp->h_cls_ptr[q];
	{
		std::vector<int32_t> fill(sp->h_cls_ptr.begin(), sp->h_cls_ptr.end() - 1);
		for(int64_t q = 0; q < ns; ++ q)
			sp->h_level_fronts[fill[level[q] * NCLS + front_cls[q]] ++] = (int32_t)q;
	}
	{
		int64_t max_steps = 0;
		for(int64_t q = 0; q < ns; ++ q)
			if(front_cls[q] == 4)
				max_steps = std::max<int64_t>(max_steps, (front_w[q] + front_pad[q]) / DENSE_NB);
		if(max_steps > 0) // (a tree without big fronts never calls the dense factor: no streams, no block inverses)
			dense_reserve(ctx, max_steps);
	}
	// dependency-driven part: every level below the first one that holds a big front the launch cannot take; with teams
	// (SPP_SPARSE_TEAMS, default on) it takes the big fronts as well -- G consecutive workgroups each -- and covers the
	// whole tree
	std::vector<int32_t> dag_list, dag_list_bwd, dag_rank, front_team(ns, 1);
	std::vector<int64_t> front_tinv(ns, 0);
	int64_t tinv_doubles = 0;
	{
		bool teams = true;
		if(const char *e = getenv("SPP_SPARSE_TEAMS"))
			teams = atoi(e) != 0;
		// a team's counter barrier needs ALL of its members resident at once (one 1024-thread workgroup with ~137 KB of LDS
		// per CU): never more members than half the CUs of this device (a partitioned or CU-masked device), and no teams
		// at all on one that cannot hold a team of two beside the rest of the launch
		hipDeviceProp_t prop;
		SPP_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
		const int team_cap = prop.multiProcessorCount / 2;
		if(team_cap < 2)
			teams = false;
		int32_t limit = (int32_t)sp->n_levels;
		if(!teams)
			for(int64_t q = 0; q < ns; ++ q)
				if(front_cls[q] == 4)
					limit = std::min(limit, level[q]);
		sp->dag_level_limit = limit;
		int max_cls = 0;
		std::vector<int32_t> fronts_in;
		for(int64_t q = 0; q < ns; ++ q)
			if(level[q] < limit) {
				fronts_in.push_back((int32_t)q);
				max_cls = std::max(max_cls, front_cls[q]);
			}
		// children first; inside a level the long fronts (large classes) first
		std::stable_sort(fronts_in.begin(), fronts_in.end(), [&](int32_t a, int32_t b) {
			return level[a] != level[b] ? level[a] < level[b] : front_cls[a] > front_cls[b]; });
		dag_list_bwd.assign(fronts_in.rbegin(), fronts_in.rend()); // parents first
		int team_max = 40, team_cols = 16; // measured on sphere2500: 12 / 64 -> 1.55 ms, 24 / 32 -> 1.34, 32 / 16 -> 1.27, 48 / 16 -> 1.26, 64 / 8 -> 1.36
		if(const char *e = getenv("SPP_SPARSE_TEAM_MAX"))
			team_max = std::max(1, std::min(64, atoi(e)));
		team_max = std::max(2, std::min(team_max, team_cap));
		if(const char *e = getenv("SPP_SPARSE_TEAM_COLS")) // columns of the padded front per member
			team_cols = std::max(8, std::min(256, atoi(e)));
		for(size_t i = 0; i < fronts_in.size(); ++ i) {
			const int32_t q = fronts_in[i];
			int G = 1;
			if(front_cls[q] == 4) {
				const int32_t hp = front_h[q] + front_pad[q];
				G = std::max(2, std::min(team_max, (hp + team_cols - 1) / team_cols)); // about one workgroup per team_cols columns of the padded front
				front_team[q] = G;
				front_tinv[q] = tinv_doubles;
				tinv_doubles += (int64_t)((front_w[q] + front_pad[q]) / DENSE_NB) * DENSE_NB * DENSE_NB;
			}
			for(int r = 0; r < G; ++ r) {
				dag_list.push_back(q);
				dag_rank.push_back(r);
			}
		}
		sp->dag_first1 = 0;
		while(sp->dag_first1 < (int32_t)dag_list.size() && level[dag_list[sp->dag_first1]] == 0)
			++ sp->dag_first1;
		sp->dag_n = (int32_t)dag_list.size();
		sp->dag_n_bwd = (int32_t)dag_list_bwd.size();
		{
			int32_t l1 = limit;
			for(size_t i = 0; i < fronts_in.size(); ++ i)
				if(front_cls[fronts_in[i]] >= 2)
					l1 = std::min(l1, level[fronts_in[i]]);
			sp->dag_split_level = l1;
			sp->dag_split = 0;
			int low_cls = 0;
			while(sp->dag_split < (int32_t)dag_list.size() && level[dag_list[sp->dag_split]] < l1) {
				low_cls = std::max(low_cls, front_cls[dag_list[sp->dag_split]]);
				++ sp->dag_split;
			}
			const size_t hpl = low_cls == 0 ? 32 : 64;
			sp->dag_lds_low = (hpl * (hpl + 1) + hpl + 2 * 16 * PT + 8) * sizeof(double);
		}
		const size_t hp = max_cls == 0 ? 32 : (max_cls == 1 ? 64 : (max_cls == 2 ? 128 : MID_FRONT_MAX));
		size_t lds_doubles = (max_cls >= 3 ? std::max<size_t>((size_t)17 * MID_FRONT_MAX + MID_FRONT_MAX, (size_t)128 * 129 + 128) : hp * (hp + 1) + hp)
			+ 2 * 16 * PT + 8;
		if(max_cls == 4)
			lds_doubles = std::max(lds_doubles, TEAM_LDS_DOUBLES);
		sp->dag_lds = lds_doubles * sizeof(double);
	}
	sp->h_front_ld = front_ld;
	sp->h_front_pad = front_pad;
	sp->h_front_cls = front_cls;
	sp->front_doubles = foff;
	sp->vbuf_doubles = voff;
	sp->h_front_h = front_h;
	sp->h_front_w = front_w;
	sp->h_front_off = front_off;
	ctx->factor_flops = (int64_t)flops;
	ctx->factor_nnz = nnz_r;
	ctx->solve_bytes = 8 * (st.nvals + nnz_r) + 16 * nnz_r + 16 * st.n; // factor + two tri-solves

	// children lists and relative index maps
	std::vector<int32_t> child_ptr(ns + 1, 0), child_list(ns), rel_ptr(ns + 1, 0), rel;
	for(int64_t s = 0; s < ns; ++ s)
		if(sn_parent[s] >= 0)
			++ child_ptr[sn_parent[s] + 1];
	for(int64_t s = 0; s < ns; ++ s)
		child_ptr[s + 1] += child_ptr[s];
	{
		std::vector<int32_t> fill(child_ptr.begin(), child_ptr.end() - 1);
		for(int64_t s = 0; s < ns; ++ s)
			if(sn_parent[s] >= 0)
				child_list[fill[sn_parent[s]] ++] = (int32_t)s;
	}
	child_list.resize(child_ptr[ns]);
	sp->h_child_ptr = child_ptr;
	sp->h_child_list = child_list;
	for(int64_t s = 0; s < ns; ++ s) {
		rel_ptr[s + 1] = rel_ptr[s];
		const int32_t p = sn_parent[s];
		if(p < 0)
			continue;
		const std::vector<int32_t> &pr = sn_rows[p];
		size_t qp = 0;
		for(size_t q = sn_ncols[s]; q < sn_rows[s].size(); ++ q) {
			const int32_t g = sn_rows[s][q];
			while(qp < pr.size() && pr[qp] < g)
				++ qp;
			SPP_REQUIRE(qp < pr.size() && pr[qp] == g, SPP_E_HIP, "internal: child row missing in parent front");
			for(int32_t e = 0; e < pdim[g]; ++ e)
				rel.push_back(loc_off[p][qp] + e);
		}
		rel.push_back(loc_off[p][pr.size()]); // the child's right-hand-side slot -> the parent's
		rel_ptr[s + 1] = (int32_t)rel.size();
	}

	// assembly lists: every stored block of Lambda goes to the front owning its (permuted) row
	std::vector<std::vector<int64_t> > a_src(ns);
	std::vector<std::vector<int32_t> > a_dst(ns), a_shape(ns);
	for(int64_t j = 0; j < nb; ++ j)
		for(int64_t p = st.col_ptr[j]; p < st.col_ptr[j + 1]; ++ p) {
			const int64_t i = st.row_idx[p];
			int64_t a = inv[i], b = inv[j];
			int tr = 0;
			if(a > b) {
				std::swap(a, b);
				tr = 1; // the stored block is (i, j) = (b, a) in permuted terms: transpose into (a, b)
			}
			const int32_t s = sn_of[a];
			const std::vector<int32_t> &r = sn_rows[s];
			const size_t qa = (size_t)(a - sn_first[s]);
			const size_t qb = (size_t)(std::lower_bound(r.begin(), r.end(), (int32_t)b) - r.begin());
			SPP_REQUIRE(qb < r.size() && r[qb] == b, SPP_E_HIP, "internal: block outside of its front");
			a_src[s].push_back((st.blk_off[p] << 1) | tr);
			a_dst[s].push_back(loc_off[s][qa] | (loc_off[s][qb] << 16));
			// shape of the destination sub-block (rows x cols)
			a_shape[s].push_back(pdim[a] | (pdim[b] << 8));
		}
	std::vector<int32_t> asm_ptr(ns + 1, 0), asm_dst, asm_shape;
	std::vector<int64_t> asm_src;
	for(int64_t s = 0; s < ns; ++ s) {
		asm_ptr[s + 1] = asm_ptr[s] + (int32_t)a_src[s].size();
		asm_src.insert(asm_src.end(), a_src[s].begin(), a_src[s].end());
		asm_dst.insert(asm_dst.end(), a_dst[s].begin(), a_dst[s].end());
		asm_shape.insert(asm_shape.end(), a_shape[s].begin(), a_shape[s].end());
	}
	sp->h_asm_ptr = asm_ptr;
	std::vector<int32_t> perm_scalar(st.n);
	for(int64_t k = 0; k < nb; ++ k)
		for(int32_t e = 0; e < pdim[k]; ++ e)
			perm_scalar[pbase[k] + e] = (int32_t)(st.base[order[k]] + e);

	if(getenv("SPP_VERBOSE")) {
		int64_t hist[8] = {0}; // h <= 16, 32, 64, 128, 256, 512, 1024, more
		double fl_hist[8] = {0};
		int32_t hmax = 0, wmax = 0;
		for(int64_t q = 0; q < ns; ++ q) {
			int b = 0;
			while(b < 7 && front_h[q] > (16 << b))
				++ b;
			++ hist[b];
			double f = 0;
			for(int32_t j = 0; j < front_w[q]; ++ j)
				f += (double)(front_h[q] - j) * (double)(front_h[q] - j);
			fl_hist[b] += f;
			hmax = std::max(hmax, front_h[q]);
			wmax = std::max(wmax, front_w[q]);
		}
		fprintf(stderr, "[spp] sparse analyze: nb %ld n %ld supernodes %ld levels %ld nnz(R) %ld flops %.3g front MB %.1f hmax %d wmax %d\n",
			(long)nb, (long)st.n, (long)ns, (long)sp->n_levels, (long)nnz_r, flops, foff * 8e-6, hmax, wmax);
		for(int b = 0; b < 8; ++ b)
			fprintf(stderr, "[spp]   fronts with h <= %4d: %6ld  flops %.3g\n", 16 << b, (long)hist[b], fl_hist[b]);
		for(int64_t l = 0; l < sp->n_levels; ++ l)
			if(l < 4 || l + 12 >= sp->n_levels) {
				int32_t hm = 0;
				for(int32_t q = sp->h_level_ptr[l]; q < sp->h_level_ptr[l + 1]; ++ q)
					hm = std::max(hm, front_h[sp->h_level_fronts[q]]);
				fprintf(stderr, "[spp]   level %3ld: %5d fronts, max h %d\n", (long)l, sp->h_level_ptr[l + 1] - sp->h_level_ptr[l], hm);
			}
	}
	hipStream_t s = ctx->stream;
	UploadArena arena(s);
	arena.add(sp->level_fronts, sp->h_level_fronts);
	arena.add(sp->front_off, front_off);
	arena.add(sp->front_h, front_h);
	arena.add(sp->front_w, front_w);
	arena.add(sp->front_ld, front_ld);
	arena.add(sp->front_pad, front_pad);
	arena.add(sp->front_voff, front_voff);
	arena.add(sp->asm_ptr, asm_ptr);
	arena.add(sp->asm_src, asm_src);
	arena.add(sp->asm_dst, asm_dst);
	arena.add(sp->asm_shape, asm_shape);
	arena.add(sp->child_ptr, child_ptr);
	arena.add(sp->child_list, child_list);
	arena.add(sp->rel_ptr, rel_ptr);
	arena.add(sp->rel, rel);
	arena.add(sp->rows_ptr, rows_ptr);
	arena.add(sp->rows, rows);
	arena.add(sp->perm_scalar, perm_scalar);
	arena.add(sp->front_cls, front_cls);
	arena.add(sp->front_parent, sn_parent);
	arena.add(sp->front_level, level);
	arena.add(sp->dag_list, dag_list);
	arena.add(sp->dag_list_bwd, dag_list_bwd);
	arena.add(sp->dag_rank, dag_rank);
	arena.add(sp->front_team, front_team);
	arena.add(sp->front_tinv, front_tinv);
	arena.commit(sp->index_store);
	sp->team_tinv.reserve((size_t)std::max<int64_t>(tinv_doubles, 1));
	sp->team_bar.reserve((size_t)std::max<int64_t>(ns, 1));
	SPP_HIP_CHECK(hipMemsetAsync(sp->team_bar.p, 0, (size_t)std::max<int64_t>(ns, 1) * sizeof(int), s));
	sp->dag_solves = 0;
	sp->dag_done.reserve((size_t)std::max<int64_t>(ns, 1));
	SPP_HIP_CHECK(hipMemsetAsync(sp->dag_done.p, 0, (size_t)std::max<int64_t>(ns, 1) * sizeof(int), s));
	sp->dag_epoch = 0;
	sp->fronts.reserve((size_t)std::max<int64_t>(foff, 2));
	sp->vbuf.reserve((size_t)std::max<int64_t>(voff, 1));
	sp->xperm.reserve((size_t)st.n);
	clk.lap("maps built, uploads enqueued");
	SPP_HIP_CHECK(hipStreamSynchronize(s));
	clk.lap("uploads done");
}

// --------------------------------------------------------------------------------------------------
// numeric kernels
// --------------------------------------------------------------------------------------------------
constexpr int FT = 256;

// local (unpadded) front index -> index in a padded image: `pad` entries are inserted after the w pivots
__device__ __forceinline__ int padded(int r, int w, int pad) { return r < w ? r : r + pad; }

// One workgroup per front, the whole front in LDS (image HP x HP, column stride HP + 1, pivot block
// padded with identity to a multiple of 16). Blocked right-looking partial factorization, 16-wide
// panels: A  16 x 16 diagonal tile factored + inverted in registers by wave 0 (diag_tile_factor)
//         B  row panel X = Dinv^T Y, one 16 x 16 MFMA tile per wave
//         C  trailing update T[I,K] -= P_I^T P_K on MFMA f64 16x16x4
// The result is written back to the front's HBM buffer in the plain (unpadded) layout.
// the arrays of the plan the frontal kernels read (one struct: the same argument block for every kernel)
struct FrontArgs {
	const int64_t *front_off;
	const int32_t *front_h, *front_w, *front_ld, *front_pad;
	const int32_t *asm_ptr;
	const int64_t *asm_src;
	const int32_t *asm_dst, *asm_shape, *child_ptr, *child_list, *rel_ptr, *rel, *rows_ptr, *rows;
	const int64_t *front_voff;
	double *xperm;
	const double *vals;
	double *fronts, *vbuf;
	int *info;
	long long *trace; // debugging: stamps 4 (extend-add done), 5 (factored), 6 (written back) of the block's eight
};

struct NoWait { __device__ __forceinline__ void operator()() const { } };

// `wait_children` is called (by every thread) once the parts of the assembly that do not depend on the children --
// clearing, identity padding, the blocks of Lambda, the right-hand side -- are done, right before the extend-add
template <int HP, int NTH, bool GMEM, class WaitFn>
__device__ __forceinline__ void front_body(const int s, const FrontArgs &fa, double *fsm, const WaitFn &wait_children)
{
	const int64_t *__restrict__ front_off = fa.front_off;
	const int32_t *__restrict__ front_h = fa.front_h, *__restrict__ front_w = fa.front_w, *__restrict__ front_ld = fa.front_ld;
	const int32_t *__restrict__ front_pad = fa.front_pad, *__restrict__ asm_ptr = fa.asm_ptr;
	const int64_t *__restrict__ asm_src = fa.asm_src;
	const int32_t *__restrict__ asm_dst = fa.asm_dst, *__restrict__ asm_shape = fa.asm_shape, *__restrict__ child_ptr = fa.child_ptr;
	const int32_t *__restrict__ child_list = fa.child_list, *__restrict__ rel_ptr = fa.rel_ptr, *__restrict__ rel = fa.rel;
	const int32_t *__restrict__ rows_ptr = fa.rows_ptr, *__restrict__ rows = fa.rows;
	const double *xperm = fa.xperm;
	const double *__restrict__ vals = fa.vals;
	double *fronts = fa.fronts;
	int *info = fa.info;
	// GMEM: the image is the front's own HBM buffer (already in the 16-padded layout, stride ld);
	// otherwise an LDS image of HP x (HP + 1) doubles
	constexpr int NW = NTH / 64;
	const int h = front_h[s], w = front_w[s], ld = front_ld[s];
	const int w16 = (w + 15) & ~15, pad = w16 - w, hp = h + pad, nt = (hp + 15) >> 4;
	double *F = fronts + front_off[s];
	const int TSF = GMEM ? ld : HP + 1;
	double *T = GMEM ? F : fsm;                                   // image
	double *Dv = GMEM ? fsm : fsm + HP * (HP + 1);                // Dinv of the current diagonal tile
	double *Gd = Dv + 16 * PT;          // (unused by the fronts, written by diag_tile_factor)
	double *dinv = Gd + 16 * PT;        // HP
	int *fail = (int*)(dinv + HP);
	// GMEM: LDS copy of the current 16-row panel (all columns of the front), stride PLS per column: the
	// trailing update takes its operands from here instead of re-reading them from HBM tile by tile
	constexpr int PLS = 17;
	double *Pl = dinv + HP + 2;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int l15 = lane & 15, l4 = lane >> 4;
	if(GMEM) {
		for(int64_t e = tid; e < (int64_t)ld * (nt * 16); e += NTH)
			T[e] = 0.0;
	} else {
		for(int e = tid; e < HP * (HP + 1); e += NTH)
			T[e] = 0.0;
	}
	if(tid == 0)
		*fail = 0;
	__syncthreads();
	if(tid >= w && tid < w16)
		T[tid + (int64_t)tid * TSF] = 1.0; // identity padding of the pivot block
	// ---- blocks of Lambda
	for(int q = asm_ptr[s] + wave; q < asm_ptr[s + 1]; q += NW) {
		const int64_t so = asm_src[q];
		const double *src = vals + (so >> 1);
		const int dr = padded(asm_dst[q] & 0xffff, w, pad), dc = padded(asm_dst[q] >> 16, w, pad);
		const int nr = asm_shape[q] & 0xff, ncol = asm_shape[q] >> 8;
		if(lane < nr * ncol) {
			const int r = lane % nr, c = lane / nr;
			T[(dr + r) + (dc + c) * TSF] = (so & 1) ? src[c + ncol * r] : src[r + nr * c];
		}
	}
	// ---- the right-hand side on the pivot rows: column h - 1 (the slot every front carries, see sparse_analyze)
	{
		const int32_t *rw = rows + rows_ptr[s];
		const int cs = padded(h - 1, w, pad);
		for(int r = tid; r < w; r += NTH)
			T[r + (int64_t)cs * TSF] = xperm[rw[r]];
	}
	wait_children();
	__syncthreads();
	// ---- extend-add of the children's update matrices (their right-hand-side column included), children in list order
	for(int cq = child_ptr[s]; cq < child_ptr[s + 1]; ++ cq) {
		const int c = child_list[cq];
		const int hc = front_h[c], wc = front_w[c], ldc = front_ld[c], oc = wc + front_pad[c];
		const double *Fc = fronts + front_off[c];
		const int32_t *rl = rel + rel_ptr[c];
		const int m = hc - wc;
		for(int e = tid; e < m * m; e += NTH) {
			const int i = e % m, j = e / m;
			if(i <= j)
				T[padded(rl[i], w, pad) + padded(rl[j], w, pad) * TSF] += Fc[(oc + i) + (int64_t)(oc + j) * ldc];
		}
		__syncthreads();
	}
	if(fa.trace && tid == 0)
		fa.trace[8 * blockIdx.x + 4] = wall_clock64();
	// ---- blocked partial factorization
	const int npan = w16 >> 4;
	for(int J = 0; J < npan; ++ J) {
		const int j0 = J * 16;
		if(wave == 0)
			diag_tile_factor_sel(TSF, T, Dv, Gd, dinv, j0, lane, fail, info, 0);
		__syncthreads();
		if(*fail)
			return;
		for(int K = J + 1 + wave; K < nt; K += NW) { // B: row panel
			double *Y = T + j0 + (K * 16) * TSF;
			const v4f64 x = tile_atb(Dv, 1, PT, Y, 1, TSF, lane);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				Y[(l4 + 4 * r) + l15 * TSF] = x[r];
			if(GMEM) {
#pragma unroll
				for(int r = 0; r < 4; ++ r)
					Pl[(l4 + 4 * r) + (K * 16 + l15) * PLS] = x[r];
			}
		}
		__syncthreads();
		if(GMEM) {
			// C for the HBM image: operands from the LDS panel copy, MFMA operands swapped so that a lane
			// group covers 16 consecutive ROWS of the target tile (128-byte segments of 4 columns per access
			// instead of 32 bytes out of 16 different cache lines), two tiles per wave in flight
			const int nI = nt - 1 - J, nR = nI * (nI + 1) / 2;
			auto decode = [&](int q, int &I, int &K) {
				int a = 0, rem = q;
				while(rem >= nI - a) {
					rem -= nI - a;
					++ a;
				}
				I = J + 1 + a;
				K = I + rem;
			};
			for(int q = wave; q < nR; q += 2 * NW) {
				const int q1 = q + NW;
				int I0, K0, I1 = 0, K1 = 0;
				decode(q, I0, K0);
				const bool have1 = q1 < nR;
				if(have1)
					decode(q1, I1, K1);
				// lane (l15, l4 + 4 r) <-> D[I 16 + l15][K 16 + l4 + 4 r]
				double *D0 = T + (I0 * 16 + l15) + (int64_t)(K0 * 16 + l4) * TSF;
				if(have1) {
					double *D1 = T + (I1 * 16 + l15) + (int64_t)(K1 * 16 + l4) * TSF;
					double o0[4], o1[4];
#pragma unroll
					for(int r = 0; r < 4; ++ r) {
						o0[r] = D0[(int64_t)(4 * r) * TSF];
						o1[r] = D1[(int64_t)(4 * r) * TSF];
					}
					v4f64 d0, d1;
					tile_atb2_rt(PLS, Pl + (K0 * 16) * PLS, Pl + (I0 * 16) * PLS, 1, PLS,
						Pl + (K1 * 16) * PLS, Pl + (I1 * 16) * PLS, 1, PLS, lane, d0, d1);
#pragma unroll
					for(int r = 0; r < 4; ++ r) {
						D0[(int64_t)(4 * r) * TSF] = o0[r] - d0[r];
						D1[(int64_t)(4 * r) * TSF] = o1[r] - d1[r];
					}
				} else {
					const v4f64 d = tile_atb(Pl + (K0 * 16) * PLS, 1, PLS, Pl + (I0 * 16) * PLS, 1, PLS, lane);
#pragma unroll
					for(int r = 0; r < 4; ++ r)
						D0[(int64_t)(4 * r) * TSF] -= d[r];
				}
			}
		} else
		{ // C: trailing update, tiles (I <= K) of the remaining (nt - J - 1) tile rows. A wave works on two
		  // tiles at a time and fetches the old target tiles together with the operands: every access is
		  // a full memory round trip when the image lives in HBM, and one tile at a time left the wave
		  // waiting on three dependent ones per tile.
			const int nI = nt - 1 - J, nR = nI * (nI + 1) / 2;
			auto decode = [&](int q, int &I, int &K) {
				int a = 0, rem = q;
				while(rem >= nI - a) {
					rem -= nI - a;
					++ a;
				}
				I = J + 1 + a;
				K = I + rem;
			};
			for(int q = wave; q < nR; q += 2 * NW) {
				const int q1 = q + NW;
				int I0, K0, I1 = 0, K1 = 0;
				decode(q, I0, K0);
				const bool have1 = q1 < nR;
				if(have1)
					decode(q1, I1, K1);
				double *D0 = T + (I0 * 16) + (K0 * 16) * TSF;
				if(have1) {
					double *D1 = T + (I1 * 16) + (K1 * 16) * TSF;
					double o0[4], o1[4];
#pragma unroll
					for(int r = 0; r < 4; ++ r) {
						o0[r] = D0[(l4 + 4 * r) + l15 * TSF];
						o1[r] = D1[(l4 + 4 * r) + l15 * TSF];
					}
					v4f64 d0, d1;
					tile_atb2_rt(TSF, T + j0 + (I0 * 16) * TSF, T + j0 + (K0 * 16) * TSF, 1, TSF,
						T + j0 + (I1 * 16) * TSF, T + j0 + (K1 * 16) * TSF, 1, TSF, lane, d0, d1);
#pragma unroll
					for(int r = 0; r < 4; ++ r) {
						D0[(l4 + 4 * r) + l15 * TSF] = o0[r] - d0[r];
						D1[(l4 + 4 * r) + l15 * TSF] = o1[r] - d1[r];
					}
				} else {
					const v4f64 d = tile_atb(T + j0 + (I0 * 16) * TSF, 1, TSF, T + j0 + (K0 * 16) * TSF, 1, TSF, lane);
#pragma unroll
					for(int r = 0; r < 4; ++ r)
						D0[(l4 + 4 * r) + l15 * TSF] -= d[r];
				}
			}
		}
		__syncthreads();
	}
	if(fa.trace && tid == 0)
		fa.trace[8 * blockIdx.x + 5] = wall_clock64();
	if(GMEM)
		return; // factored in place
	// ---- write back the upper triangle in the plain layout
	for(int e = tid; e < h * h; e += NTH) {
		const int r = e % h, c = e / h;
		if(r <= c)
			F[r + (int64_t)c * ld] = T[padded(r, w, pad) + padded(c, w, pad) * TSF];
	}
	if(fa.trace && tid == 0)
		fa.trace[8 * blockIdx.x + 6] = wall_clock64();
}

template <int HP, int NTH, bool GMEM>
__global__ __launch_bounds__(NTH)
void front_lds_kernel(const int32_t *__restrict__ list, FrontArgs fa)
{
	extern __shared__ double fsm[];
	front_body<HP, NTH, GMEM>(list[blockIdx.x], fa, fsm, NoWait());
}

// ---- big fronts (image does not fit LDS): assembled in HBM in the padded layout, factored by the
// multi-workgroup dense kernels (spp_dense.hip). ONE launch assembles a front: a workgroup owns BFC columns of the
// padded front and, for them, clears the column, sets the identity padding of the pivot block, takes the pivot rows of
// the right-hand side (slot column), the blocks of Lambda, and then the update matrices of the children IN LIST ORDER
// (column ownership: no races, the same summation order run to run). It used to be a memset, a scatter kernel and one
// extend-add kernel per child -- five launches of ~5 us each in front of a 28 us factorization.
constexpr int BFC = 8;

__global__ __launch_bounds__(256)
void bigfront_assemble_kernel(int s, const int64_t *__restrict__ front_off, const int32_t *__restrict__ front_h,
	const int32_t *__restrict__ front_w, const int32_t *__restrict__ front_ld, const int32_t *__restrict__ front_pad,
	const int32_t *__restrict__ asm_ptr, const int64_t *__restrict__ asm_src, const int32_t *__restrict__ asm_dst,
	const int32_t *__restrict__ asm_shape, const int32_t *__restrict__ child_ptr, const int32_t *__restrict__ child_list,
	const int32_t *__restrict__ rel_ptr, const int32_t *__restrict__ rel, const int32_t *__restrict__ rows_ptr,
	const int32_t *__restrict__ rows, const double *__restrict__ xperm, const double *__restrict__ vals,
	double *__restrict__ fronts)
{
	__shared__ int jrange[2];
	const int h = front_h[s], w = front_w[s], ld = front_ld[s], pad = front_pad[s];
	const int hp = h + pad;
	double *F = fronts + front_off[s];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int c0 = (int)blockIdx.x * BFC, c1 = (c0 + BFC < hp) ? c0 + BFC : hp; // padded columns of this workgroup
	if(c0 >= hp)
		return;
	for(int e = tid; e < (c1 - c0) * ld; e += 256)
		F[(int64_t)c0 * ld + e] = 0.0;
	__syncthreads();
	for(int i = c0 + tid; i < c1; i += 256)
		if(i >= w && i < w + pad)
			F[i + (int64_t)i * ld] = 1.0; // identity padding of the pivot block
	{
		const int cs = padded(h - 1, w, pad); // the right-hand-side slot
		if(cs >= c0 && cs < c1) {
			const int32_t *rw = rows + rows_ptr[s];
			for(int r = tid; r < w; r += 256)
				F[r + (int64_t)cs * ld] = xperm[rw[r]];
		}
	}
	// ---- blocks of Lambda: every workgroup walks the front's list and keeps the elements of its columns
	for(int q = asm_ptr[s] + wave; q < asm_ptr[s + 1]; q += 4) {
		const int64_t so = asm_src[q];
		const double *src = vals + (so >> 1);
		const int dr = padded(asm_dst[q] & 0xffff, w, pad), dc = padded(asm_dst[q] >> 16, w, pad);
		const int nr = asm_shape[q] & 0xff, ncol = asm_shape[q] >> 8;
		if(dc + ncol <= c0 || dc >= c1)
			continue; // wave-uniform
		if(lane < nr * ncol) {
			const int r = lane % nr, c = lane / nr;
			if(dc + c >= c0 && dc + c < c1)
				F[(dr + r) + (int64_t)(dc + c) * ld] = (so & 1) ? src[c + ncol * r] : src[r + nr * c];
		}
	}
	__syncthreads();
	// ---- extend-add, children in list order; a child's rows map to ascending parent indices, so the child columns
	// that land in [c0, c1) are one contiguous range [ja, jb)
	for(int cq = child_ptr[s]; cq < child_ptr[s + 1]; ++ cq) {
		const int c = child_list[cq];
		const int hc = front_h[c], wc = front_w[c], ldc = front_ld[c], oc = wc + front_pad[c];
		const double *Fc = fronts + front_off[c];
		const int32_t *rl = rel + rel_ptr[c];
		const int m = hc - wc;
		if(tid == 0) {
			int lo = 0, hi = m; // first j with padded(rl[j]) >= c0
			while(lo < hi) {
				const int mid = (lo + hi) >> 1;
				if(padded(rl[mid], w, pad) < c0) lo = mid + 1; else hi = mid;
			}
			jrange[0] = lo;
			hi = m; // first j with padded(rl[j]) >= c1
			while(lo < hi) {
				const int mid = (lo + hi) >> 1;
				if(padded(rl[mid], w, pad) < c1) lo = mid + 1; else hi = mid;
			}
			jrange[1] = lo;
		}
		__syncthreads();
		const int ja = jrange[0], jb = jrange[1];
		for(int e = tid; e < (jb - ja) * m; e += 256) {
			const int j = ja + e / m, i = e % m;
			if(i <= j)
				F[padded(rl[i], w, pad) + (int64_t)padded(rl[j], w, pad) * ld] += Fc[(oc + i) + (int64_t)(oc + j) * ldc];
		}
		__syncthreads();
	}
}

// ---- multifrontal triangular solves ------------------------------------------------------------------
// forward  R^T y = b (leaves -> root):  no kernel of its own -- b rides through the factorization as the slot column
//          of every front (sparse_analyze), which leaves y on the pivot rows of that column.
// backward R x = y (root -> leaves), blocked by 64 pivots:    per block  t_B = v_B - sum_{c after B} R[B, c] x_c  (lane = row of
//          the block: every column contributes 64 contiguous doubles), then x_B = R_BB^-1 t_B.
// v = work vector of the front (length h, unpadded local indices) in HBM.
constexpr int SB = 64;

constexpr int BWD_VL = 1024; // fronts up to this height keep their work vector in LDS

// `wait_parent` is called (by every thread) once what does not depend on the ancestors -- y on the pivot rows, the
// triangle of the first block -- has been fetched
// NT threads; a wave takes U columns of the part right of a block per round (U loads in flight per lane)
template <int NT, int U, class WaitFn>
__device__ __forceinline__ void front_bwd_body(const int s, const FrontArgs &fa, double *tri, double (*part)[SB], double *vl,
	const WaitFn &wait_parent)
{
	const int64_t *__restrict__ front_off = fa.front_off;
	const int32_t *__restrict__ front_h = fa.front_h, *__restrict__ front_w = fa.front_w, *__restrict__ front_ld = fa.front_ld;
	const int32_t *__restrict__ front_pad = fa.front_pad;
	const int64_t *__restrict__ front_voff = fa.front_voff;
	const int32_t *__restrict__ rows_ptr = fa.rows_ptr, *__restrict__ rows = fa.rows;
	const double *fronts = fa.fronts;
	double *vbuf = fa.vbuf, *xperm = fa.xperm;
	const int h = front_h[s], w = front_w[s], ld = front_ld[s], pad = front_pad[s];
	const double *F = fronts + front_off[s];
	double *v = (h <= BWD_VL) ? vl : vbuf + front_voff[s];
	const int32_t *rw = rows + rows_ptr[s];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int nblk = (w + SB - 1) / SB;
	auto load_tri = [&](int bk) {
		const int j0 = bk * SB;
		const int nbk = (w - j0 < SB) ? (w - j0) : SB;
		for(int e = tid; e < nbk * nbk; e += NT) {
			const int i = e % nbk, k = e / nbk;
			if(i <= k)
				tri[i + k * (SB + 1)] = F[(j0 + i) + (int64_t)(j0 + k) * ld];
		}
	};
	// own part: y = the slot column of the factored front (the forward substitution happened inside the factorization)
	{
		const int cs = padded(h - 1, w, pad);
		for(int c = tid; c < w; c += NT)
			v[c] = F[c + (int64_t)cs * ld];
	}
	if(nblk > 0)
		load_tri(nblk - 1);
	wait_parent();
	// beyond: final x of the ancestors; the slot itself takes no part (0)
	for(int c = w + tid; c < h; c += NT)
		v[c] = (c < h - 1) ? xperm[rw[c]] : 0.0;
	__syncthreads();
	for(int bk = nblk; bk > 0;) {
		-- bk;
		const int j0 = bk * SB;
		const int nbk = (w - j0 < SB) ? (w - j0) : SB;
		if(bk != nblk - 1)
			load_tri(bk);
		// lane = row j0 + lane of the block; the waves split the columns after the block
		{
			double acc[U];
#pragma unroll
			for(int u = 0; u < U; ++ u)
				acc[u] = 0;
			const double *rowp = F + j0 + lane;
			int c = j0 + nbk + wave * U;
			if(lane < nbk) {
				for(; c + U <= h; c += U * (NT / 64)) {
					double f[U];
#pragma unroll
					for(int u = 0; u < U; ++ u)
						f[u] = rowp[(int64_t)padded(c + u, w, pad) * ld];
#pragma unroll
					for(int u = 0; u < U; ++ u)
						acc[u] += f[u] * v[c + u];
				}
				for(int u = 0; u < U && c + u < h; ++ u) // ragged last group (belongs to exactly one wave)
					acc[0] += rowp[(int64_t)padded(c + u, w, pad) * ld] * v[c + u];
			}
			double sum = 0;
#pragma unroll
			for(int u = 0; u < U; u += 4)
				sum += (acc[u] + acc[u + 1]) + (acc[u + 2] + acc[u + 3]);
			part[wave][lane] = sum;
		}
		__syncthreads();
		if(wave == 0) {
			double t = 0;
			if(lane < nbk) {
				t = v[j0 + lane];
#pragma unroll
				for(int q = 0; q < NT / 64; ++ q)
					t -= part[q][lane];
			}
			const double rinv = (lane < nbk) ? 1.0 / tri[lane + lane * (SB + 1)] : 0.0;
#pragma unroll
			for(int i = SB; i > 0;) { // reciprocals of the diagonal once, the running value through v_readlane with compile-time lanes
				-- i;
				if(i < nbk) { // wave-uniform
					const double xi = readlane_f64(t, i) * readlane_f64(rinv, i);
					const double tcol = tri[lane + i * (SB + 1)];
					t = (lane == i) ? xi : ((lane < i) ? t - tcol * xi : t);
				}
			}
			if(lane < nbk)
				v[j0 + lane] = t;
		}
		__syncthreads();
	}
	for(int c = tid; c < w; c += NT)
		xperm[rw[c]] = v[c];
}

template <int NT, int U>
__global__ __launch_bounds__(NT)
void front_bwd_kernel(const int32_t *__restrict__ level_fronts, FrontArgs fa)
{
	__shared__ double tri[SB * (SB + 1)];
	__shared__ double part[NT / 64][SB];
	__shared__ double vl[BWD_VL];
	front_bwd_body<NT, U>(level_fronts[blockIdx.x], fa, tri, part, vl, NoWait());
}

// --------------------------------------------------------------------------------------------------
// Dependency-driven launches. Level by level the factorization of a pose graph is 13-21 levels x up to three size
// classes = 30-60 launches of 13-17 us each, every one as long as its slowest front and separated by a launch
// boundary, plus one launch per level for the backward substitution. Here ONE launch covers all those levels: a
// workgroup per front, enumerated children first (level order), which waits for its children's flags in device
// memory (their update matrices are extend-added by the parent), runs the class's kernel body and publishes its own
// flag behind an agent-scope release. A front starts the moment its own children are done; the critical path is the
// heaviest root-to-leaf chain of fronts, not the sum over the levels of the slowest front of each.
// Progress: workgroups are dispatched in block order, so every producer a resident workgroup waits for is resident or
// finished. Every wait is bounded all the same (wall clock -> abort word -> every later wait falls through -> the host
// reports the failure and goes back to the level-by-level launches).
// Flags hold the epoch of the solve that last finished the front: no clearing between solves.
// --------------------------------------------------------------------------------------------------
struct DagArgs {
	const int32_t *list;      // fronts in dispatch order
	const int32_t *front_cls; // size class per front
	const int32_t *front_parent, *front_level;
	int *done;                // [ns] epoch of the last completed factorization / substitution of the front
	int epoch;
	int level_limit;          // fronts of levels >= level_limit are handled by launches of their own (no flags)
	int level_first;          // factorization: the fronts of levels < level_first were finished by earlier launches
	int *abort;
	long long timeout_ticks;
	// big fronts: a team of workgroups each (consecutive blocks)
	const int32_t *rank;       // per block: rank inside its front's team
	const int32_t *front_team; // per front: team size (1: a single workgroup)
	int *team_bar;             // per front: monotonic barrier counter
	const int64_t *front_tinv; // per front: offset of its diagonal-block inverses in tinv
	double *tinv;
	int solve_index;           // number of factorizations run on these counters before this one
	long long *trace;          // debugging (SPP_DAG_TRACE): per block wall-clock stamps start / children done / end
};

__device__ __forceinline__ bool dag_wait(const int *flag, int value, int *abort, long long timeout_ticks)
{
	const long long t0 = wall_clock64();
	for(int it = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value; ++ it) {
		if((it & 15) == 15) {
			if(__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
				return false;
			if(wall_clock64() - t0 > timeout_ticks) {
				__hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return false;
			}
		}
		__builtin_amdgcn_s_sleep(1);
	}
	return true;
}

// the workgroup's stores are complete and visible device-wide, then the flag
__device__ __forceinline__ void dag_publish(int *flag, int value)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if(threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// ---- big fronts inside the dependency-driven launch: a TEAM of workgroups per front -------------------------------
// Host-driven, a big front (image in HBM, pivot block padded to a multiple of 128) cost a launch for its assembly and
// three to five per 128 pivots, one front after the other (sphere2500: twelve of them, 0.95 of 1.5 ms). Here G
// consecutive workgroups of the launch share the front: each assembles a contiguous range of columns (the children
// in list order: deterministic), then per 128 pivots rank 0 factors the diagonal block in LDS (potrf_diag_body), all
// solve the row panel in 16-column slabs and update the trailing part in 64 x 64 blocks (the staged MFMA tile
// products of the dense factor), with a barrier of the team -- a monotonic counter in device memory behind an
// agent-scope release, an acquire behind the wait -- between the phases. Fronts of one level run side by side, the
// part of the assembly that needs no child runs ahead of the wait for the children.
__device__ __forceinline__ void team_barrier(int *counter, int target, int *abort, long long timeout_ticks)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if(threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const long long t0 = wall_clock64();
		for(int it = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++ it) {
			if((it & 15) == 15) {
				if(__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
					break;
				if(wall_clock64() - t0 > timeout_ticks) {
					__hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					break;
				}
			}
			__builtin_amdgcn_s_sleep(1);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	__syncthreads();
}


// the three heavy phases as functions of their own: inlined into the team body their register pressure adds to the
// state the body keeps across them and the allocator (128 VGPRs at 1024 threads) spills inside the hot loops
__device__ __noinline__ void team_potrf(double *Ablk, int64_t ld, int n_valid, double *tv, int *info, int64_t k0, double *sm)
{
	potrf_diag_body<false, 0, 1>(Ablk, ld, n_valid, 0, tv, info, k0, sm);
}

__device__ __noinline__ void team_panel_tile(int64_t n0, int64_t nright, const double *tv, double *P, int64_t ld, double *sm)
{
	panel_solve_slab<TEAM_THREADS>(n0, nright, tv, P, ld, sm);
}

__device__ __noinline__ void team_update_tile(int64_t m0, int64_t n0, int64_t nright, const double *P, int64_t ld, double *C, double *sm)
{
	gemm_tn_staged_tile<64, 64, 16, 16, 0, 0>(m0, n0, nright, nright, P, ld, P, ld, C, ld, sm);
}

template <class WaitFn>
__device__ __forceinline__ void bigfront_team_body(const int s, const int rank, const int G, const FrontArgs &fa,
	int *bar, const int bar_base, double *tinv, int *abort, long long timeout_ticks, double *sm, const WaitFn &wait_children)
{
	__shared__ int jrange[2];
	const int h = fa.front_h[s], w = fa.front_w[s], ld = fa.front_ld[s], pad = fa.front_pad[s];
	const int hp = h + pad;
	double *F = fa.fronts + fa.front_off[s];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// ---- assembly of the columns [c0, c1) of the padded front
	const int per = (((hp + G - 1) / G) + 7) & ~7;
	const int c0 = (rank * per < hp) ? rank * per : hp, c1 = (c0 + per < hp) ? c0 + per : hp;
	for(int64_t e = tid; e < (int64_t)(c1 - c0) * ld; e += TEAM_THREADS)
		F[(int64_t)c0 * ld + e] = 0.0;
	__syncthreads();
	for(int i = c0 + tid; i < c1; i += TEAM_THREADS)
		if(i >= w && i < w + pad)
			F[i + (int64_t)i * ld] = 1.0; // identity padding of the pivot block
	{
		const int cs = padded(h - 1, w, pad); // the right-hand-side slot
		if(cs >= c0 && cs < c1) {
			const int32_t *rw = fa.rows + fa.rows_ptr[s];
			for(int r = tid; r < w; r += TEAM_THREADS)
				F[r + (int64_t)cs * ld] = fa.xperm[rw[r]];
		}
	}
	// blocks of Lambda: every workgroup walks the front's list and keeps the elements of its columns
	for(int q = fa.asm_ptr[s] + wave; q < fa.asm_ptr[s + 1]; q += TEAM_THREADS / 64) {
		const int64_t so = fa.asm_src[q];
		const double *src = fa.vals + (so >> 1);
		const int dr = padded(fa.asm_dst[q] & 0xffff, w, pad), dc = padded(fa.asm_dst[q] >> 16, w, pad);
		const int nr = fa.asm_shape[q] & 0xff, ncol = fa.asm_shape[q] >> 8;
		if(dc + ncol <= c0 || dc >= c1)
			continue; // wave-uniform
		if(lane < nr * ncol) {
			const int r = lane % nr, c = lane / nr;
			if(dc + c >= c0 && dc + c < c1)
				F[(dr + r) + (int64_t)(dc + c) * ld] = (so & 1) ? src[c + ncol * r] : src[r + nr * c];
		}
	}
	wait_children();
	__syncthreads();
	// extend-add, children in list order; a child's rows map to ascending parent indices, so the child columns that
	// land in [c0, c1) are one contiguous range [ja, jb)
	for(int cq = fa.child_ptr[s]; cq < fa.child_ptr[s + 1]; ++ cq) {
		const int c = fa.child_list[cq];
		const int hc = fa.front_h[c], wc = fa.front_w[c], ldc = fa.front_ld[c], oc = wc + fa.front_pad[c];
		const double *Fc = fa.fronts + fa.front_off[c];
		const int32_t *rl = fa.rel + fa.rel_ptr[c];
		const int m = hc - wc;
		if(tid == 0) {
			int lo = 0, hi = m; // first j with padded(rl[j]) >= c0
			while(lo < hi) {
				const int mid = (lo + hi) >> 1;
				if(padded(rl[mid], w, pad) < c0) lo = mid + 1; else hi = mid;
			}
			jrange[0] = lo;
			hi = m; // first j with padded(rl[j]) >= c1
			while(lo < hi) {
				const int mid = (lo + hi) >> 1;
				if(padded(rl[mid], w, pad) < c1) lo = mid + 1; else hi = mid;
			}
			jrange[1] = lo;
		}
		__syncthreads();
		const int ja = jrange[0], jb = jrange[1];
		for(int e = tid; e < (jb - ja) * m; e += TEAM_THREADS) {
			const int j = ja + e / m, i = e % m;
			if(i <= j)
				F[padded(rl[i], w, pad) + (int64_t)padded(rl[j], w, pad) * ld] += Fc[(oc + i) + (int64_t)(oc + j) * ldc];
		}
		__syncthreads();
	}
	int nbar = 0;
	team_barrier(bar, bar_base + G * (++ nbar), abort, timeout_ticks);
	// ---- partial factorization, 128 pivots per step
	const int nsteps = (w + pad) / NB;
	for(int j = 0; j < nsteps; ++ j) {
		const int k0 = j * NB, k1 = k0 + NB;
		double *tv = tinv + (size_t)j * NB * NB;
		if(rank == 0) {
			int n_valid = w - k0;
			n_valid = n_valid < 0 ? 0 : (n_valid > NB ? NB : n_valid);
			team_potrf(F + k0 + (int64_t)k0 * ld, ld, n_valid, tv, fa.info, (int64_t)k0, sm);
		}
		team_barrier(bar, bar_base + G * (++ nbar), abort, timeout_ticks);
		const int nright = hp - k1; // columns right of the diagonal block (the slot column included)
		double *P = F + k0 + (int64_t)k1 * ld;
		// row panel R_kj = (R_kk^-1)^T F_kj in place, 16 columns per tile (eight waves; the other eight idle)
		for(int sl = rank; sl * 16 < nright; sl += G) {
			team_panel_tile((int64_t)sl * 16, nright, tv, P, ld, sm);
			__syncthreads(); // the LDS panels are staged anew by the next tile
		}
		team_barrier(bar, bar_base + G * (++ nbar), abort, timeout_ticks);
		// trailing update F[k1.., k1..] -= P^T P, upper 64 x 64 blocks
		{
			const int nb64 = (nright + 63) >> 6, ntile = nb64 * (nb64 + 1) / 2;
			double *C = F + k1 + (int64_t)k1 * ld;
			for(int t = rank; t < ntile; t += G) {
				int bj = 0, rem = t; // tile t -> (bi <= bj), block columns first
				while(rem > bj) {
					rem -= bj + 1;
					++ bj;
				}
				const int bi = rem;
				team_update_tile((int64_t)bi * 64, (int64_t)bj * 64, nright, P, ld, C, sm);
				__syncthreads();
			}
		}
		team_barrier(bar, bar_base + G * (++ nbar), abort, timeout_ticks);
	}
}

constexpr int DAG_THREADS = 1024;

__global__ __launch_bounds__(DAG_THREADS)
void front_dag_kernel(DagArgs da, FrontArgs fa)
{
	extern __shared__ double fsm[];
	const int s = da.list[blockIdx.x];
	const int cls = da.front_cls[s];
	const int nth = cls == 0 ? 64 : (cls == 1 ? 256 : (cls == 2 ? 512 : 1024));
	const int tid = threadIdx.x;
	if(tid >= nth)
		return; // (a barrier does not wait for waves that have ended)
	if(da.trace && tid == 0)
		da.trace[8 * blockIdx.x] = wall_clock64();
	if(cls == 4) {
		const int G = da.front_team[s], rank = da.rank[blockIdx.x];
		auto wait_kids = [&]() {
			if(tid == 0) {
				if(da.trace)
					da.trace[8 * blockIdx.x + 1] = wall_clock64();
				for(int cq = fa.child_ptr[s]; cq < fa.child_ptr[s + 1]; ++ cq) {
					const int c = fa.child_list[cq];
					if(da.front_level[c] >= da.level_first && !dag_wait(da.done + c, da.epoch, da.abort, da.timeout_ticks))
						break;
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if(da.trace)
					da.trace[8 * blockIdx.x + 2] = wall_clock64();
			}
		};
		const int nbar = 1 + 3 * ((fa.front_w[s] + fa.front_pad[s]) / NB);
		bigfront_team_body(s, rank, G, fa, da.team_bar + s, da.solve_index * G * nbar, da.tinv + da.front_tinv[s], da.abort,
			da.timeout_ticks, fsm, wait_kids);
		if(rank == 0) // (every member released its stores in the last barrier of the team)
			dag_publish(da.done + s, da.epoch);
		if(da.trace && tid == 0)
			da.trace[8 * blockIdx.x + 3] = wall_clock64();
		return;
	}
	// the children's flags are awaited inside the body, after the part of the assembly that does not need them
	auto wait_children = [&]() {
		if(tid == 0) {
			if(da.trace)
				da.trace[8 * blockIdx.x + 1] = wall_clock64();
			for(int cq = fa.child_ptr[s]; cq < fa.child_ptr[s + 1]; ++ cq) {
				const int c = fa.child_list[cq];
				if(da.front_level[c] >= da.level_first && !dag_wait(da.done + c, da.epoch, da.abort, da.timeout_ticks))
					break; // aborted: the front is computed from garbage and discarded by the host
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if(da.trace)
				da.trace[8 * blockIdx.x + 2] = wall_clock64();
		}
	};
	if(cls == 0)
		front_body<32, 64, false>(s, fa, fsm, wait_children);
	else if(cls == 1)
		front_body<64, 256, false>(s, fa, fsm, wait_children);
	else if(cls == 2)
		front_body<128, 512, false>(s, fa, fsm, wait_children);
	else
		front_body<MID_FRONT_MAX, 1024, true>(s, fa, fsm, wait_children);
	dag_publish(da.done + s, da.epoch);
	if(da.trace && tid == 0)
		da.trace[8 * blockIdx.x + 3] = wall_clock64();
}

__global__ __launch_bounds__(1024)
void front_bwd_dag_kernel(DagArgs da, FrontArgs fa)
{
	__shared__ double tri[SB * (SB + 1)];
	__shared__ double part[1024 / 64][SB];
	__shared__ double vl[BWD_VL];
	const int s = da.list[blockIdx.x];
	// wide fronts (hundreds of columns right of a 64-pivot block): 16 waves, eight loads in flight per lane; the others: 4 waves
	const bool wide = fa.front_h[s] > 192;
	if(!wide && threadIdx.x >= FT)
		return; // (a barrier does not wait for waves that have ended)
	auto wait_parent = [&]() {
		if(threadIdx.x == 0) {
			const int p = da.front_parent[s];
			if(p >= 0 && da.front_level[p] < da.level_limit) // (ancestors above the limit were finished by earlier launches)
				dag_wait(da.done + p, da.epoch, da.abort, da.timeout_ticks);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__syncthreads();
	};
	if(wide)
		front_bwd_body<1024, 8>(s, fa, tri, part, vl, wait_parent);
	else
		front_bwd_body<FT, 4>(s, fa, tri, part, vl, wait_parent);
	dag_publish(da.done + s, da.epoch);
}

__global__ void gather_perm_kernel(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ src,
	double *__restrict__ dst)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n)
		dst[i] = src[perm[i]];
}

// `abort` (nullable): the word a timed-out flag wait of the dependency-driven launches sets -- the right-hand side is then
// left as it came in, and the host repeats the solve level by level from the intact inputs
__global__ void scatter_perm_kernel(int64_t n, const int32_t *__restrict__ perm, const double *__restrict__ src,
	double *__restrict__ dst, const int *__restrict__ abort)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(abort && *abort)
		return;
	if(i < n)
		dst[perm[i]] = src[i];
}

static FrontArgs make_front_args(spp_ctx *ctx, SparsePlan *sp, const double *d_vals)
{
	FrontArgs fa;
	fa.front_off = sp->front_off.p;
	fa.front_h = sp->front_h.p; fa.front_w = sp->front_w.p; fa.front_ld = sp->front_ld.p; fa.front_pad = sp->front_pad.p;
	fa.asm_ptr = sp->asm_ptr.p; fa.asm_src = sp->asm_src.p; fa.asm_dst = sp->asm_dst.p; fa.asm_shape = sp->asm_shape.p;
	fa.child_ptr = sp->child_ptr.p; fa.child_list = sp->child_list.p; fa.rel_ptr = sp->rel_ptr.p; fa.rel = sp->rel.p;
	fa.rows_ptr = sp->rows_ptr.p; fa.rows = sp->rows.p;
	fa.front_voff = sp->front_voff.p;
	fa.xperm = sp->xperm.p;
	fa.vals = d_vals;
	fa.fronts = sp->fronts.p;
	fa.vbuf = sp->vbuf.p;
	fa.info = ctx->dense.info.p;
	fa.trace = nullptr;
	return fa;
}

template <int HP, int NTH, bool GMEM>
static void launch_front_lds(spp_ctx *ctx, SparsePlan *sp, int32_t b, int32_t e, const double *d_vals)
{
	if(e <= b)
		return;
	const size_t lds = ((GMEM ? (size_t)17 * HP : (size_t)HP * (HP + 1)) + 2 * 16 * PT + HP + 8) * sizeof(double);
	static uint64_t attr_seen = 0;
	if(first_on_this_device(attr_seen)) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)front_lds_kernel<HP, NTH, GMEM>,
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	}
	hipLaunchKernelGGL((front_lds_kernel<HP, NTH, GMEM>), dim3((unsigned)(e - b)), dim3(NTH), lds, ctx->stream,
		sp->level_fronts.p + b, make_front_args(ctx, sp, d_vals));
}

static void sparse_enqueue(spp_ctx *ctx, const double *d_vals, double *d_rhs);

// (A hipGraph replay of the whole solve was measured at 1 - 2.5 % per solve in round 1 and cannot carry the dependency-
// driven launches -- epochs and counters are kernel arguments --: removed.)
int sparse_factor_solve(spp_ctx *ctx, const double *d_vals, double *d_rhs)
{
	SparsePlan *sp = ctx->sparse;
	SPP_REQUIRE(sp, SPP_E_STATE, "sparse plan missing");
	sparse_enqueue(ctx, d_vals, d_rhs);
	// the status is fetched after the solves (no host round trip between factorization and solves)
	bool dag_aborted = false;
	int info = dense_info_fetch(ctx, &dag_aborted);
	if(dag_aborted) {
		// a front of the dependency-driven launches timed out waiting for another front's flag (a device that does not
		// keep the launch's workgroups resident): the right-hand side was left untouched, the plan now launches level by
		// level -- the same solve again, from the intact inputs
		sparse_enqueue(ctx, d_vals, d_rhs);
		info = dense_info_fetch(ctx);
	}
	return info ? SPP_NOT_POSDEF : SPP_OK;
}

static void sparse_enqueue(spp_ctx *ctx, const double *d_vals, double *d_rhs)
{
	SparsePlan *sp = ctx->sparse;
	hipStream_t s = ctx->stream;
	const unsigned gn = (unsigned)((sp->n + 255) / 256);
	dense_info_reset(ctx);
	phase_begin(ctx, SPP_PHASE_FACTOR);
	// P b: every front takes its pivot rows of it as its right-hand-side column
	hipLaunchKernelGGL(gather_perm_kernel, dim3(gn), dim3(256), 0, s, sp->n, sp->perm_scalar.p, d_rhs, sp->xperm.p);
	static int use_dag = -1;
	if(use_dag < 0) {
		const char *e = getenv("SPP_SPARSE_DAG"); // 0: one launch per level and size class (round 1 / 2 schedule)
		use_dag = e ? atoi(e) : 1;
	}
	const bool dag = use_dag && sp->dag_ok && sp->dag_n > 0;
	const int32_t first_level = dag ? sp->dag_level_limit : 0;
	DagArgs da;
	da.front_cls = sp->front_cls.p;
	da.front_parent = sp->front_parent.p;
	da.front_level = sp->front_level.p;
	da.done = sp->dag_done.p;
	da.level_limit = sp->dag_level_limit;
	da.abort = ctx->dense.info.p + 3;
	static long long dag_timeout = -1;
	if(dag_timeout < 0) {
		const char *e = getenv("SPP_DAG_TIMEOUT_TICKS"); // debugging / tests: a tiny value forces the timeout fallback
		dag_timeout = e ? atoll(e) : (long long)(500.0 * 1e5); // 500 ms of the 100 MHz wall clock
	}
	da.timeout_ticks = dag_timeout;
	da.epoch = 0;
	da.level_first = 0;
	da.list = nullptr;
	da.rank = sp->dag_rank.p;
	da.front_team = sp->front_team.p;
	da.team_bar = sp->team_bar.p;
	da.front_tinv = sp->front_tinv.p;
	da.tinv = sp->team_tinv.p;
	da.solve_index = 0;
	da.trace = nullptr;
	if(dag) {
		static uint64_t attr_seen = 0;
		if(first_on_this_device(attr_seen)) {
			SPP_HIP_CHECK(hipFuncSetAttribute((const void*)front_dag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
				(int)(std::max<size_t>(std::max<size_t>((size_t)17 * MID_FRONT_MAX + MID_FRONT_MAX, (size_t)128 * 129 + 128) + 2 * 16 * PT + 8, TEAM_LDS_DOUBLES) * sizeof(double))));
		}
		if(sp->dag_solves >= (1 << 18) || sp->dag_epoch >= (1 << 30)) {
			// the teams' barrier counters and the fronts' epochs are monotonic 32-bit words (solve_index * G * barriers per
			// solve stays below 2^31 for 2^18 solves of the largest team): start over long before they wrap
			SPP_HIP_CHECK(hipMemsetAsync(sp->team_bar.p, 0, sp->team_bar.cap * sizeof(int), s));
			SPP_HIP_CHECK(hipMemsetAsync(sp->dag_done.p, 0, sp->dag_done.cap * sizeof(int), s));
			sp->dag_solves = 0;
			sp->dag_epoch = 0;
		}
		da.epoch = ++ sp->dag_epoch;
		da.solve_index = sp->dag_solves ++;
		// (the leaves as plain launches of their own size class in front of this one -- the dependency-driven launch reserves
		// the LDS of the largest class for every workgroup -- measured no gain on either pose graph: dropped)
		const int32_t skip = 0;
		da.list = sp->dag_list.p + skip;
		da.rank = sp->dag_rank.p + skip;
		static int trace_left = -1;
		if(trace_left < 0)
			trace_left = getenv("SPP_DAG_TRACE") ? atoi(getenv("SPP_DAG_TRACE")) : 0;
		DevBuf<long long> trace_buf;
		if(trace_left > 0) {
			trace_buf.reserve((size_t)8 * sp->dag_n);
			SPP_HIP_CHECK(hipMemsetAsync(trace_buf.p, 0, (size_t)8 * sp->dag_n * sizeof(long long), s));
			da.trace = trace_buf.p;
		}
		static int split_env = -1;
		if(split_env < 0)
			split_env = getenv("SPP_DAG_SPLIT") ? atoi(getenv("SPP_DAG_SPLIT")) : 1;
		if(sp->dag_n > skip) {
			FrontArgs fa = make_front_args(ctx, sp, d_vals);
			fa.trace = da.trace;
			if(split_env && !da.trace && sp->dag_split >= (split_env >= 2 ? 1 : 1024) && sp->dag_split < sp->dag_n) { // (SPP_DAG_SPLIT=2: whatever the size)
				// the bottom of the tree (fronts of at most 64 rows): 256 threads and their own LDS, four workgroups per CU;
				// the launch above it skips the waits for those children (finished by the kernel boundary)
				hipLaunchKernelGGL(front_dag_kernel, dim3((unsigned)sp->dag_split), dim3(256), sp->dag_lds_low, s, da, fa);
				DagArgs db = da;
				db.list = sp->dag_list.p + sp->dag_split;
				db.rank = sp->dag_rank.p + sp->dag_split;
				db.level_first = sp->dag_split_level;
				hipLaunchKernelGGL(front_dag_kernel, dim3((unsigned)(sp->dag_n - sp->dag_split)), dim3(DAG_THREADS), sp->dag_lds, s, db, fa);
			} else
				hipLaunchKernelGGL(front_dag_kernel, dim3((unsigned)(sp->dag_n - skip)), dim3(DAG_THREADS), sp->dag_lds, s, da, fa);
		}
		if(trace_left > 0 && -- trace_left == 0) { // debugging: per level, when its fronts started / had their children / ended (us)
			std::vector<long long> tr((size_t)8 * sp->dag_n);
			SPP_HIP_CHECK(hipMemcpyAsync(tr.data(), trace_buf.p, tr.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
			SPP_HIP_CHECK(hipStreamSynchronize(s));
			std::vector<int32_t> lst((size_t)sp->dag_n), lev((size_t)sp->n_snodes), cl((size_t)sp->n_snodes);
			SPP_HIP_CHECK(hipMemcpy(lst.data(), sp->dag_list.p, lst.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
			SPP_HIP_CHECK(hipMemcpy(lev.data(), sp->front_level.p, lev.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
			long long t0 = tr[0];
			for(size_t b = 0; b < lst.size(); ++ b)
				if(tr[8 * b])
					t0 = std::min(t0, tr[8 * b]);
			for(int64_t l = 0; l < sp->n_levels; ++ l) {
				double s0 = 1e30, s1 = 0, k1 = 0, e0 = 1e30, e1 = 0, dmax = 0, ph[4] = {0, 0, 0, 0};
				int cnt = 0, hmax = 0;
				for(size_t b = 0; b < lst.size(); ++ b) {
					const int32_t f = lst[b];
					if(lev[f] != l || !tr[8 * b])
						continue;
					++ cnt;
					const double ts = (tr[8 * b] - t0) * 0.01, tk = (tr[8 * b + 2] - t0) * 0.01, te = (tr[8 * b + 3] - t0) * 0.01;
					s0 = std::min(s0, ts); s1 = std::max(s1, ts); k1 = std::max(k1, tk); e0 = std::min(e0, te); e1 = std::max(e1, te);
					if(te - tk > dmax && tr[8 * b + 4]) { // phases of the front with the longest body: extend-add, factor, write-back, publish
						ph[0] = (tr[8 * b + 4] - tr[8 * b + 2]) * 0.01;
						ph[1] = (tr[8 * b + 5] - tr[8 * b + 4]) * 0.01;
						ph[2] = tr[8 * b + 6] ? (tr[8 * b + 6] - tr[8 * b + 5]) * 0.01 : 0;
						ph[3] = (tr[8 * b + 3] - (tr[8 * b + 6] ? tr[8 * b + 6] : tr[8 * b + 5])) * 0.01;
					}
					dmax = std::max(dmax, te - tk);
					hmax = std::max(hmax, sp->h_front_h[f]);
				}
				fprintf(stderr, "[spp dag] level %2ld: %4d blocks, max h %4d, start %.1f..%.1f, last children-done %.1f, end %.1f..%.1f, longest body after children %.1f us (extend-add %.1f, factor %.1f, write-back %.1f, publish %.1f)\n",
					(long)l, cnt, hmax, s0, s1, k1, e0, e1, dmax, ph[0], ph[1], ph[2], ph[3]);
			}
		}
		da.trace = nullptr;
	}
	for(int64_t l = first_level; l < sp->n_levels; ++ l) {
		const int32_t *cp = sp->h_cls_ptr.data() + l * NCLS;
		launch_front_lds<32, 64, false>(ctx, sp, cp[0], cp[1], d_vals);
		launch_front_lds<64, 256, false>(ctx, sp, cp[1], cp[2], d_vals);
		launch_front_lds<128, 512, false>(ctx, sp, cp[2], cp[3], d_vals);
		launch_front_lds<MID_FRONT_MAX, 1024, true>(ctx, sp, cp[3], cp[4], d_vals); // in place in HBM
		for(int32_t q = cp[4]; q < cp[5]; ++ q) { // big fronts, one after the other
			const int32_t f = sp->h_level_fronts[q];
			const int32_t h = sp->h_front_h[f], w = sp->h_front_w[f], pad = sp->h_front_pad[f], ld = sp->h_front_ld[f];
			const int32_t hp = h + pad;
			double *F = sp->fronts.p + sp->h_front_off[f];
			hipLaunchKernelGGL(bigfront_assemble_kernel, dim3((unsigned)((hp + BFC - 1) / BFC)), dim3(256), 0, s,
				f, sp->front_off.p, sp->front_h.p, sp->front_w.p, sp->front_ld.p, sp->front_pad.p, sp->asm_ptr.p, sp->asm_src.p,
				sp->asm_dst.p, sp->asm_shape.p, sp->child_ptr.p, sp->child_list.p, sp->rel_ptr.p, sp->rel.p, sp->rows_ptr.p,
				sp->rows.p, sp->xperm.p, d_vals, sp->fronts.p);
			struct IdentGuard { // pivots [w, w + pad) are identity padding: the diagonal-block kernel skips their panels
				DenseWork &d;
				IdentGuard(DenseWork &dw, int64_t from) : d(dw) { d.ident_from = from; }
				~IdentGuard() { d.ident_from = -1; }
			} guard(ctx->dense, w);
			dense_factor_steps(ctx, F, ld, w + pad, hp, hp, (w + pad) / DENSE_NB, false);
		}
	}
	phase_end(ctx, SPP_PHASE_FACTOR);
	SPP_HIP_CHECK(hipGetLastError());
	phase_begin(ctx, SPP_PHASE_TRISOLVE);
	// the forward substitution R^T y = P b was carried by the factorization (slot column of every front): backward only
	for(int64_t l = sp->n_levels; l > first_level;) {
		-- l;
		const int32_t b = sp->h_level_ptr[l], e = sp->h_level_ptr[l + 1];
		// the levels up here hold the wide fronts (hundreds of columns right of a block): 16 waves, 8 loads in flight each
		if(l >= sp->dag_level_limit)
			hipLaunchKernelGGL((front_bwd_kernel<1024, 8>), dim3((unsigned)(e - b)), dim3(1024), 0, s,
				sp->level_fronts.p + b, make_front_args(ctx, sp, d_vals));
		else
			hipLaunchKernelGGL((front_bwd_kernel<FT, 4>), dim3((unsigned)(e - b)), dim3(FT), 0, s,
				sp->level_fronts.p + b, make_front_args(ctx, sp, d_vals));
	}
	if(dag) { // the levels below: one launch, a front waits for its parent
		da.epoch = ++ sp->dag_epoch;
		da.list = sp->dag_list_bwd.p;
		hipLaunchKernelGGL(front_bwd_dag_kernel, dim3((unsigned)sp->dag_n_bwd), dim3(1024), 0, s, da, make_front_args(ctx, sp, d_vals));
	}
	hipLaunchKernelGGL(scatter_perm_kernel, dim3(gn), dim3(256), 0, s, sp->n, sp->perm_scalar.p, sp->xperm.p, d_rhs,
		dag ? ctx->dense.info.p + 3 : nullptr);
	phase_end(ctx, SPP_PHASE_TRISOLVE);
	SPP_HIP_CHECK(hipGetLastError());
}

} // namespace spp
