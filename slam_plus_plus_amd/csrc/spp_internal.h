// spp_internal.h -- private declarations shared by the translation units of libspp_hip.so.
// MI355X (gfx950) only: no CUDA shims, no CPU fallback. See include/spp_hip.h for the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <utility>
#include <new>
#include <exception>
#include <thread>
#include <stdexcept>
#include <algorithm>
#include <string.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/spp_hip.h"

namespace spp {

// ------------------------------------------------------------------------------------------------
// errors: internal code throws, the extern "C" layer converts to SPP_E_* codes
// ------------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
	int code;
	Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

#define SPP_HIP_CHECK(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) \
	throw spp::Error(SPP_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); } while(0)
#define SPP_REQUIRE(cond, code, msg) do { if(!(cond)) throw spp::Error((code), (msg)); } while(0)

// hipFuncSetAttribute is per DEVICE: true the first time the calling site runs on the current device (a process may hold
// contexts on several GPUs)
inline bool first_on_this_device(uint64_t &seen)
{
	int dev = 0;
	(void)hipGetDevice(&dev);
	const uint64_t bit = uint64_t(1) << (dev & 63);
	if(seen & bit)
		return false;
	seen |= bit;
	return true;
}

// wall clock of the host-side phases of an analysis, printed lap by lap when SPP_VERBOSE is set
struct VClock {
	const char *who;
	std::chrono::steady_clock::time_point t;
	bool on;
	explicit VClock(const char *w) : who(w), t(std::chrono::steady_clock::now()), on(getenv("SPP_VERBOSE") != nullptr) {}
	void lap(const char *what)
	{
		if(!on)
			return;
		const std::chrono::steady_clock::time_point n = std::chrono::steady_clock::now();
		fprintf(stderr, "[spp] %s: %-34s %8.2f ms\n", who, what, std::chrono::duration<double>(n - t).count() * 1e3);
		t = n;
	}
};

// ------------------------------------------------------------------------------------------------
// device buffer (owned, grows geometrically like the reference's workspaces,
// LinearSolver_UberBlock.h:332-348)
// ------------------------------------------------------------------------------------------------
// Host arrays of the symbolic phases: tens of MB each, written once front to back. Two costs of a plain std::vector
// dominate them -- the value-initialization of resize() (a serial memset) and the first touch in 4 KB pages (63 000
// page faults per 260 MB) -- so big blocks come 2 MB-aligned and offered to transparent huge pages, and elements of
// trivial type are default-initialized (NOT zeroed: every user writes before it reads).
template <class T>
struct HugeAlloc {
	typedef T value_type;
	HugeAlloc() {}
	template <class U> HugeAlloc(const HugeAlloc<U>&) {}
	T *allocate(size_t n)
	{
		const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
		void *q = nullptr;
		if(bytes >= ((size_t)4 << 20)) {
			if(posix_memalign(&q, huge, (bytes + huge - 1) & ~(huge - 1)) != 0)
				q = nullptr;
#ifdef MADV_HUGEPAGE
			if(q)
				madvise(q, bytes, MADV_HUGEPAGE);
#endif
		}
		if(!q)
			q = malloc(bytes ? bytes : 1);
		if(!q)
			throw std::bad_alloc();
		return (T*)q;
	}
	void deallocate(T *p, size_t) { free(p); }
	template <class U> void construct(U *p) { ::new((void*)p) U; } // default-init
	template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&... a) { ::new((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
	template <class U> bool operator==(const HugeAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <class T> using HVec = std::vector<T, HugeAlloc<T> >;

template <class T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0; // elements
	bool owned = true; // false: a view into an UploadArena's single allocation
	DevBuf() {}
	DevBuf(const DevBuf&) = delete;
	DevBuf &operator=(const DevBuf&) = delete;
	~DevBuf() { release(); }
	void release()
	{
		if(p && owned)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
		owned = true;
	}
	void view(T *ptr, size_t n)
	{
		release();
		p = ptr;
		cap = n;
		owned = false;
	}
	void reserve(size_t n)
	{
		if(n <= cap && owned)
			return;
		release();
		size_t want = n;
		hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
		if(e != hipSuccess) {
			p = nullptr;
			throw Error(SPP_E_NOMEM, "hipMalloc of " + std::to_string(want * sizeof(T)) + " bytes failed");
		}
		cap = want;
	}
	template <class A>
	void upload(const std::vector<T, A> &h, hipStream_t s)
	{
		reserve(h.size() ? h.size() : 1);
		if(!h.empty())
			SPP_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
	}
};

// The index arrays of a plan -- a few dozen of them, kilobytes to megabytes each -- in ONE device allocation filled by ONE
// copy: a hipMalloc + a pageable-memory hipMemcpyAsync per array was most of the analysis time of a pose graph
// (26 arrays: 10-18 ms of a 14 ms sparse_analyze). add() stages a vector, commit() uploads everything and turns the
// DevBufs into views of the allocation `store` owns; the host image must live until the stream has been synchronized.
struct UploadArena {
	hipStream_t stream;
	std::vector<unsigned char> host;
	struct Item { void **pp; size_t *pcap; bool *powned; size_t off, n; };
	std::vector<Item> items;
	explicit UploadArena(hipStream_t s) : stream(s) { host.reserve(size_t(1) << 20); }
	template <class T, class A>
	void add(DevBuf<T> &b, const std::vector<T, A> &v)
	{
		if(v.size() * sizeof(T) > (size_t(256) << 10)) { // a large array gains nothing from being staged twice: its own upload
			b.upload(v, stream);
			return;
		}
		b.release();
		const size_t off = (host.size() + 255) & ~(size_t)255, bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
		host.resize(off + bytes);
		if(!v.empty())
			memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
		Item it = {(void**)&b.p, &b.cap, &b.owned, off, std::max<size_t>(v.size(), 1)};
		items.push_back(it);
	}
	void commit(DevBuf<unsigned char> &store)
	{
		store.release();
		store.reserve(std::max<size_t>(host.size(), 256));
		if(!host.empty())
			SPP_HIP_CHECK(hipMemcpyAsync(store.p, host.data(), host.size(), hipMemcpyHostToDevice, stream));
		for(size_t i = 0; i < items.size(); ++ i) {
			*items[i].pp = store.p + items[i].off;
			*items[i].pcap = items[i].n;
			*items[i].powned = false;
		}
	}
};

// ------------------------------------------------------------------------------------------------
// host-side copy of the analyzed Lambda structure
// ------------------------------------------------------------------------------------------------
struct Structure {
	int64_t nb = 0, n = 0, nnzb = 0, nvals = 0;
	HVec<int64_t> col_ptr, row_idx, blk_off, base;
	HVec<int32_t> dim;
};

// ------------------------------------------------------------------------------------------------
// Schur plan (guided ordering: poses first, landmarks last; LinearSolver_Schur.cpp:771-838)
// ------------------------------------------------------------------------------------------------
// one work item of the S accumulation: a chunk of the pair list of one block of S, with everything the wave
// needs to finish it in ONE 32-byte (scalar) load -- the chain of dependent fetches per item (item -> block id
// -> A offset / block coordinates -> A block) used to cost six memory round trips
struct SaccItem {
	int32_t beg, end;  // pair range
	int32_t kind;      // 0: dp x dp block of the dense S at element offset dst (leading dimension ld),
	                   // 1: contiguous dp*dp values at S + dst (sparse reduced system), 2: partial slot at partial + dst
	int32_t pad;
	int64_t dst;
	int64_t aoff;      // offset of the A block in vals that this item adds, or -1
};

struct SchurPlan {
	int dp = 0, dl = 0;            // pose / landmark block width
	int64_t nc = 0, nl = 0;        // poses, landmarks owned by this shard
	int64_t nl_total = 0;
	int64_t no = 0;                // pose-landmark blocks (observations) of this shard
	int64_t n_red = 0, ld = 0;     // reduced dimension, padded leading dimension
	int64_t n_sblk = 0;            // blocks of S that are written (upper, incl. diagonal)
	int64_t n_pairs = 0;           // sum_p k_p (k_p + 1) / 2
	int64_t n_items = 0, n_multi = 0; // work items (block chunks) / blocks split over several items
	bool add_A = true;             // this shard adds A and the pose rhs (rank 0)
	bool u_landmark_major = true;  // layout of the packed U blocks (Up): observation order instead of camera-major
	bool factored = true;          // S accumulation on ONE packed block per observation, V = U F with C^-1 = F F^T (spp_schur.hip)
	// reduced camera system kept SPARSE (block-CSC, dp x dp blocks) and solved by the supernodal path:
	// S buffer = [ s_st.nvals block values | n_red reduced rhs ]
	bool sparse_S = false;
	bool mis = false;              // partition by a maximal independent set instead of by block width
	Structure s_st;
	DevBuf<int64_t> sblk_voff;     // [n_sblk] offset of the S block in the sparse value array
	// host copies needed later
	std::vector<int64_t> pose_block; // reduced pose index -> original block column
	std::vector<int64_t> lm_block;   // owned landmark index -> original block column
	std::vector<uint8_t> is_lm;      // per block column: eliminated by the Schur complement (any shard)
	// device arrays
	DevBuf<int32_t> lm_ptr;        // [nl+1] first obs of landmark
	DevBuf<int32_t> bs_ptr;        // [n_bs+1] landmark ranges of the fused back-substitution (at most 256 observations each); n_bs = 0: two launches
	int64_t n_bs = 0;
	DevBuf<int64_t> lm_coff;       // [nl] offset of C block in vals
	DevBuf<int64_t> lm_rbase;      // [nl] scalar offset of the landmark in rhs
	DevBuf<int32_t> obs_pose;      // [no] reduced pose index
	DevBuf<int32_t> obs_lm;        // [no] owned landmark index
	DevBuf<int64_t> obs_off;       // [no] (offset in vals << 1) | transposed
	DevBuf<int64_t> pose_rbase;    // [nc] scalar offset of the pose in rhs
	DevBuf<int32_t> cam_ptr;       // [nc+1] obs list per pose (ascending landmark)
	DevBuf<int32_t> cam_obs;       // [no]
	DevBuf<int32_t> obs_wpos;      // [no] position of the observation in the per-pose (camera-major) lists:
	                               //      W, Up and xw are stored camera-major (locality of the S accumulation)
	// S accumulation work items
	DevBuf<SaccItem> items;        // [n_items] (ordered by camera tiles, not by block)
	DevBuf<int32_t> xcd_beg;       // [9] item range of each XCD (equal work, not equal counts)
	int32_t xcd_max_items = 0;     // longest of those ranges
	DevBuf<int32_t> sblk_i1, sblk_i2; // [n_sblk]
	DevBuf<int64_t> sblk_aoff;     // [n_sblk] offset of the A block in vals or -1
	DevBuf<int32_t> pair_a, pair_b; // [n_pairs]
	DevBuf<int32_t> multi_blk;     // [n_multi] S block id of split blocks
	DevBuf<int32_t> multi_ptr;     // [n_multi+1] slot range
	// numeric workspaces
	DevBuf<double> cinv;           // [nl * dl*dl]   -(C^-1)
	DevBuf<double> lfac;           // [nl * dl*dl]   F = chol(C)^-T, upper triangular, C^-1 = F F^T (factored form)
	DevBuf<double> W;              // [no * dp*dl]   -U C^-1, camera-major
	DevBuf<double> Up;             // [no * dp*dl]   U packed, camera-major
	DevBuf<double> xw;             // [no * dp]      W l per observation
	DevBuf<double> partial;        // [slots * dp*dp]
	DevBuf<double> S;              // [ld*ld + ld] when the caller does not supply the buffer
	void release_all();
};

// ------------------------------------------------------------------------------------------------
// dense Cholesky workspace
// ------------------------------------------------------------------------------------------------
struct DenseWork {
	DevBuf<double> tinv;           // inverse of the current diagonal block (NB x NB)
	DevBuf<int> info;              // device flag: 0 ok, j+1 = pivot j non-positive
	DevBuf<double> tinv_all;       // inverses of all diagonal blocks (nblk x NB x NB) kept for the solves
	int64_t tinv_half = 0;         // > 0: that many of them are in the two-halves form the factorization leaves (completed before a solve)
	DevBuf<double> xtmp;           // solution of the backward substitution before it replaces y
	DevBuf<int> flags;             // per block row: epoch of the solve that last published x_b (chain kernel)
	DevBuf<int> tail_pub;          // streamed tail of the dense factor: per tile, (epoch << 4) | row tiles published
	DevBuf<double> tail_dinv;      // ... and the inverse 16 x 16 diagonal tiles it publishes
	int tail_epoch = 0;
	DevBuf<int> tail_order;        // workgroup -> tile of the streamed launch, for tail_order_tr x tail_order_tc tiles
	int tail_order_tr = 0, tail_order_tc = 0;
	int tail_rows_last = 0;        // tile rows the last factorization streamed (diagnostics: SPP_INFO_DENSE_STREAMED)
	DevBuf<double> trsv_m;         // M_b = Tinv_b R_{b, b+1} per block row (M form of the backward substitution)
	DevBuf<double> trsv_pay;       // hand-over pairs {value, check word} of the one-workgroup chain: x (nblk x 128) and w (nblk x 128)
	int epoch = 0;
	int *h_chain_err = nullptr;
	hipStream_t aux = nullptr;     // lookahead stream: potrf_diag + trsm of the next panel
	hipEvent_t ev[2] = {nullptr, nullptr};
	// device-flag hand-offs between the chain stream and the bulk stream (dense_factor_steps_enqueue):
	// sync[2 k] = row panel k complete, sync[2 k + 1] = bulk update k complete, as the epoch of the factorization
	DevBuf<int> sync;
	DevBuf<int> fuse_cnt;          // per step: sub-tiles of the next diagonal tile finished (update_potrf_kernel)
	std::vector<int> fuse_expect;  // host: the value every counter will have reached behind the launches enqueued so far
	bool fuse_dirty = false;       // counters and bookkeeping disagree (aborted factorization): cleared before the next use
	int64_t ident_from = -1;       // >= 0: the pivots from this index on are exact identity padding (set around a call by the sparse path)
	// lookahead factorization (spp_dense_la.h): persistent chain kernel on its own stream (CU-masked to the reserved CUs),
	// counters of one factorization, and whether the chain / bulk stream pair was seen to run concurrently
	hipStream_t chain = nullptr;
	hipEvent_t ev_chain = nullptr;
	DevBuf<int> la_cnt;
	DevBuf<long long> la_trace;
	int la_state = 0;              // 0: untested, 1: usable, -1: off
	int la_reserve = 0;            // CUs kept free of the bulk stream (= workgroups the chain kernel may have)
	int la_ncu = 0;
	int sync_epoch = 0;
	int sync_state = 0;            // 0: not tested on this stream, 1: the streams run concurrently, -1: disabled
	hipStream_t sync_stream = nullptr; // the ctx stream the self-test ran against
};

// ------------------------------------------------------------------------------------------------
// sparse (multifrontal supernodal) plan -- spp_sparse.hip / spp_symbolic.cpp
// ------------------------------------------------------------------------------------------------
struct SparsePlan;

// ------------------------------------------------------------------------------------------------
// assembly plan -- spp_assemble.hip
// ------------------------------------------------------------------------------------------------
struct AssemblePlan;

struct PhaseTimer {
	hipEvent_t ev[2 * SPP_N_PHASES];
	bool used[SPP_N_PHASES];
	bool created = false;
};

} // namespace spp

struct spp_ctx {
	int device = 0;
	int flags = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	std::string last_error;
	int mode = -1; // -1: not analyzed
	int shard_rank = 0, shard_world = 1;
	spp::Structure st;
	std::vector<int64_t> order; // elimination order (block columns)
	std::thread plan_trash; // releases the host image of the last Schur plan (hundreds of MB) beside the caller; joined by the next analysis and by spp_destroy
	spp::SchurPlan schur;
	spp::DenseWork dense;
	spp::SparsePlan *sparse = nullptr;
	spp::AssemblePlan *assemble = nullptr;
	// staging buffers for the host-pointer entry points
	spp::DevBuf<double> d_vals, d_rhs;
	double *h_staging = nullptr; // page-locked host buffer handed out by spp_host_staging()
	size_t h_staging_cap = 0;    // doubles
	spp::DevBuf<double> geom_partial; // partial sums of ||dx||^2 (spp_geometry.hip)
	// profiling
	spp::PhaseTimer timer;
	double phase_ms[SPP_N_PHASES] = {0};
	// dominant-kernel accounting (MFMA trailing update)
	std::vector<hipEvent_t> dom_events; // pairs
	size_t dom_used = 0;
	double dom_flops = 0;
	int64_t factor_flops = 0, solve_bytes = 0, factor_nnz = 0;
};

namespace spp {

// ---- spp_geometry.hip ----
void se2_linearize(spp_ctx *ctx, int64_t ne, const int32_t *d_v0, const int32_t *d_v1, const double *d_poses,
	const double *d_meas, double *d_J0, double *d_J1, double *d_r);
double se2_update(spp_ctx *ctx, int64_t nv, double *d_poses, const double *d_dx, bool apply);
double edge_chi2(spp_ctx *ctx, int64_t ne, int rd, const double *d_r, const double *d_Om);
void edge_robust_weights(spp_ctx *ctx, int64_t ne, int rd, int kind, double scale, double param, const double *d_r, double *d_w);
double edge_hessian_maxdiag(spp_ctx *ctx, int64_t ne, int rd, int d0, int d1, const double *d_J0, const double *d_J1,
	const double *d_Om);
double lm_gain_denominator(spp_ctx *ctx, int64_t n, const double *d_dx, const double *d_rhs, double alpha);
void se3_linearize(spp_ctx *ctx, int64_t ne, const int32_t *d_v0, const int32_t *d_v1, const double *d_poses,
	const double *d_meas, double *d_J0, double *d_J1, double *d_r);
double se3_update(spp_ctx *ctx, int64_t nv, double *d_poses, const double *d_dx, bool apply);
void ba_linearize(spp_ctx *ctx, int64_t no, const int32_t *d_cam_of, const int32_t *d_pt_of, const double *d_cams,
	const double *d_intr, const double *d_pts, const double *d_meas, double *d_J0, double *d_J1, double *d_r);
double ba_update(spp_ctx *ctx, int64_t nc, double *d_cams, const int64_t *d_cam_dxoff, int64_t np, double *d_pts,
	const int64_t *d_pt_dxoff, const double *d_dx, int64_t n_dx, bool apply);

// ---- spp_symbolic.cpp ----
void min_degree_order(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, std::vector<int64_t> &order);
void nested_dissection_order(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, std::vector<int64_t> &order);
void build_schur_plan(spp_ctx *ctx, bool sparse_S, bool mis = false);
int64_t schur_buffer_doubles(const spp_ctx *ctx); // S | rhs buffer the Schur entry points work on
bool schur_applicable(const Structure &st, int *dp, int *dl);
double schur_plan_host_probe(const Structure &st, int shard_rank, int shard_world, bool sparse_S, int64_t *out); // host only: plan + checksum, seconds

// ---- spp_sparse (symbolic on host + numeric on device) ----
void sparse_analyze(spp_ctx *ctx, const Structure &st); // plan for `st` (Lambda, or the sparse reduced system)
int sparse_factor_solve(spp_ctx *ctx, const double *d_vals, double *d_rhs);
void sparse_release(spp_ctx *ctx);
void sparse_dag_disable(spp_ctx *ctx); // after a timed-out flag wait: level-by-level launches from then on
int64_t sparse_info(const spp_ctx *ctx, int what);

// ---- spp_schur.hip ----
void schur_form(spp_ctx *ctx, const double *d_vals, const double *d_rhs, double *d_S_rhs);
int schur_finish(spp_ctx *ctx, const double *d_vals, double *d_S_rhs, double *d_rhs);
void schur_pack(spp_ctx *ctx, double *S, double *packed, bool pack);

// ---- spp_dense.hip ----
constexpr int DENSE_NB = 128;
int dense_potrf_upper(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld, bool keep_inverses);
void dense_potrf_upper_enqueue(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld);
int dense_info_fetch(spp_ctx *ctx, bool *dag_aborted = nullptr); // dag_aborted given: a timed-out sparse launch is reported there instead of thrown
void dense_potrs_upper(spp_ctx *ctx, const double *d_R, int64_t n, int64_t ld, double *d_b);
void dense_aux_park(int device, hipStream_t s); // hands the bulk stream of a closing context to the next one
void dense_factor_steps(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n, int64_t rows, int64_t ncols,
	int64_t nsteps, bool has_rhs);
void dense_info_reset(spp_ctx *ctx);
void dense_reserve(spp_ctx *ctx, int64_t nblk); // workspaces for nblk diagonal blocks (call at analyze time)

void dense_chain_check(spp_ctx *ctx); // call after the stream was synchronized
void dense_set_padding(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n);
bool dense_gemm_tn_sub(spp_ctx *ctx, int64_t m, int64_t n, int64_t k, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only);
double microbench_copy(spp_ctx *ctx, size_t bytes, int iters);
double microbench_mfma_f64(spp_ctx *ctx, int iters);
double microbench_ctile(spp_ctx *ctx, int n, int iters);
double microbench_update(spp_ctx *ctx, int64_t m, int iters);

// ---- spp_assemble.hip ----
void assemble_analyze(spp_ctx *ctx, int64_t nv, const int32_t *dim, int64_t ne, const int64_t *v0,
	const int64_t *v1, int d0, int d1, int rd, int64_t unary_vertex);
void assemble_run(spp_ctx *ctx, const double *J0, const double *J1, const double *Om, const double *r,
	double damping, double *vals, double *eta);
void assemble_set_edge_weights(spp_ctx *ctx, const double *d_w); // null: plain edges
void assemble_release(spp_ctx *ctx);
void assemble_get_structure(const spp_ctx *ctx, int64_t *col_ptr, int64_t *row_idx, int64_t *blk_off);

// ---- profiling helpers (spp_api.cpp) ----
void phase_begin(spp_ctx *ctx, int phase);
void phase_end(spp_ctx *ctx, int phase);
void phases_reset(spp_ctx *ctx);
void phases_collect(spp_ctx *ctx);
// dominant-kernel event bracket
void dom_begin(spp_ctx *ctx);
void dom_end(spp_ctx *ctx, double flops);

// ---- host threads of the symbolic phases (spp_symbolic.cpp, spp_assemble.hip)
// Runs fn(t) for t = 0 .. nt-1 on nt host threads (fn(0) on the caller's). The symbolic phase is integer work over
// tens of millions of block products; its passes are cut into independent pieces with precomputed output offsets, so
// the result does not depend on the number of threads.
template <class F>
inline void run_threads(int nt, F fn)
{
	if(nt <= 1) {
		fn(0);
		return;
	}
	std::vector<std::thread> th;
	std::exception_ptr err[64];
	for(int t = 1; t < nt; ++ t)
		th.emplace_back([&, t]() { try { fn(t); } catch(...) { err[t] = std::current_exception(); } });
	try { fn(0); } catch(...) { err[0] = std::current_exception(); }
	for(size_t t = 0; t < th.size(); ++ t)
		th[t].join();
	for(int t = 0; t < nt; ++ t)
		if(err[t])
			std::rethrow_exception(err[t]);
}

inline int plan_threads(int64_t work)
{
	static int env = -1;
	if(env < 0) {
		const char *e = getenv("SPP_PLAN_THREADS"); // host threads of the symbolic phase (default: up to 16)
		env = e ? std::max(1, atoi(e)) : 0;
	}
	static int64_t min_work = -1;
	if(min_work < 0) {
		const char *e = getenv("SPP_PLAN_MIN_WORK"); // (tests: 1 cuts even the smallest problem among the threads)
		min_work = e ? std::max<int64_t>(1, atoll(e)) : (int64_t(1) << 18);
	}
	int nt = env ? env : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
	if(work < min_work)
		nt = 1; // small problems: a thread costs more than it saves
	return std::min(nt, 64);
}

// cut [0, n) into nt pieces of about equal weight; w_prefix has n + 1 entries (w_prefix[0] = 0)
template <class V>
inline void balanced_cuts(const V &w_prefix, int nt, std::vector<int64_t> &cut)
{
	const int64_t n = (int64_t)w_prefix.size() - 1, total = w_prefix[n];
	cut.assign(nt + 1, n);
	cut[0] = 0;
	for(int t = 1; t < nt; ++ t)
		cut[t] = std::lower_bound(w_prefix.begin(), w_prefix.end(), total * t / nt) - w_prefix.begin();
	for(int t = 1; t <= nt; ++ t)
		cut[t] = std::max(cut[t], cut[t - 1]);
	cut[nt] = n;
}


} // namespace spp
