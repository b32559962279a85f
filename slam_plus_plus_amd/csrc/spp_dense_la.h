// spp_dense_la.h -- the dense factor as a LOOKAHEAD tile Cholesky: one persistent "chain" kernel on a few reserved
// compute units + one bulk trailing-update launch per step on the rest of the chip. Included by spp_dense.hip (one
// translation unit: it uses gemm_tn_tile and the staged tile product).
//
// Functionally this is still the reference's dense reduced solve (Eigen LLT, src/slam/LinearSolver_Schur.cpp:2314-2331;
// CULA spec src/slam/LinearSolver_Schur_GPU.cpp:759): R^T R = S, upper, block 128, the right-hand side riding along as a
// padding column. What changed against the two-stream schedule of rounds 1-2 is WHO does what and how they meet:
//
//   tile (i, j) = rows 128 i.., columns 128 j.. of the matrix; step k eliminates diagonal tile (k, k)
//   chain kernel (1 + G1 + G2 workgroups of 1024 threads, resident for the whole factorization)
//     workgroup 0          potrf(k): waits for the sub-tiles of (k, k), factors + inverts it in LDS (potrf_diag_body)
//     G1 "window" group    b1  R(k, k+1) = Rkk^-T S(k, k+1)         16-column slabs                       [critical]
//                          c1  S(k+1, k+1) -= R(k, k+1)^T R(k, k+1)   32 x 32 sub-tiles -> potrf(k+1)     [critical]
//                          b2  R(k, k+2);  c2  S(k+1, k+2), S(k+2, k+2) -= ...   (the rest of the 2 x 2 window)
//     G2 "panel" group     d   R(k, j), j >= k+3: the rest of row panel k, the inverse staged in LDS once per step
//   bulk kernel, one launch per step on the CU-masked stream: S(i, j) -= R(k, i)^T R(k, j) for every upper tile with
//     i >= k+1 outside the window {(k+1,k+1), (k+1,k+2), (k+2,k+2)}; the tiles of the first three tile rows come first,
//     as 64 x 64 quarters, and each signals a counter (the chain needs them one and two steps later)
//
// The critical path of a step is potrf -> b1 -> c1 -> potrf: three hand-offs through counters in device memory, no kernel
// launch, no stream event, nothing of the bulk side (the window keeps the chain two steps ahead of what it needs from the
// bulk launches). Every dependency points to a task that is EARLIER in the order (step, potrf < b1 < c1 < b2 < c2 < d <
// bulk), every executor (workgroup of the chain kernel, the bulk stream) runs its tasks in that order: the smallest
// unfinished task is always runnable, so there is no cycle. All waits are bounded by the 100 MHz wall clock (abort word ->
// every later wait falls through -> the host reports the failure and returns to the event schedule).
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup hand-offs): everything the chain kernel publishes is stored
// write-through (agent-scope atomic = sc1 stores), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup barrier,
// ONE lane adds to the counter. A consumer polls relaxed with ONE lane, that lane runs an agent-scope acquire, waits for
// it, the workgroup barrier, then plain loads. The bulk kernel publishes the tiles of its first three tile rows the same
// way (write-through stores of the 64 x 64 quarters, drained, one counter add per quarter).
#pragma once

namespace spp {

struct LaArgs {
	double *A;
	int64_t ld, n_id, rows, ncols;
	int nsteps, nt;      // nt: tile columns = ceil(ncols / 128)
	int has_rhs;
	int g1, g2;          // window group / panel group workgroups
	double *tinv_all;
	int *info;
	int *cnt;            // counters of this factorization (zeroed by a memset in front of the launch)
	int *abort;
	long long timeout_ticks;
	int bulk_acquire;    // 1: bulk workgroups run an agent-scope acquire behind their poll
	int use444;          // whole tiles of the bulk update by the v_mfma_f64_4x4x4_4b kernel (spp_dense_444.h)
	long long *trace;    // optional: wall-clock stamps (SPP_LA_TRACE)
};

// counter layout (ints), NS = nsteps, NT = nt
//   potrf_done[k]            k
//   diag_cnt[k]              NS + k            sub-tiles of (k, k) updated by c1(k - 1)           (k <= NS)
//   win_cnt[k][w]            2 NS + 1 + 2 k + w    w = 0: tile (k+1, k+2), w = 1: tile (k+2, k+2)  by c2(k)
//   col_cnt[k][j]            4 NS + 1 + k NT + j   16-column slabs of R(k, j) solved
//   bt_cnt[k][r][j]          4 NS + 1 + NS NT + (3 k + r) NT + j   quarters of bulk(k)'s tile (k+1+r, j) done, r = 0..2
//   pan_next[k]              4 NS + 1 + 4 NS NT + k    tile columns of row panel k handed out so far (panel group)
__host__ __device__ inline int la_potrf_done(const LaArgs &a, int k) { return k; }
__host__ __device__ inline int la_diag_cnt(const LaArgs &a, int k) { return a.nsteps + k; }
__host__ __device__ inline int la_win_cnt(const LaArgs &a, int k, int w) { return 2 * a.nsteps + 1 + 2 * k + w; }
__host__ __device__ inline int la_col_cnt(const LaArgs &a, int k, int j) { return 4 * a.nsteps + 1 + k * a.nt + j; }
__host__ __device__ inline int la_bt_cnt(const LaArgs &a, int k, int r, int j) { return 4 * a.nsteps + 1 + a.nsteps * a.nt + (3 * k + r) * a.nt + j; }
__host__ __device__ inline int la_pan_next(const LaArgs &a, int k) { return 4 * a.nsteps + 1 + a.nsteps * a.nt * 4 + k; } // next tile column of row panel k to hand out
__host__ __device__ inline size_t la_counter_ints(int nsteps, int nt) { return (size_t)(4 * nsteps + 1) + (size_t)nsteps * nt * 4 + nsteps + 16; }

// 16-column slabs of tile column j (0 beyond the matrix)
__host__ __device__ inline int la_nslabs(const LaArgs &a, int j)
{
	int64_t w = a.ncols - (int64_t)NB * j;
	if(w <= 0)
		return 0;
	if(w > NB)
		w = NB;
	return (int)((w + 15) >> 4);
}
// extent of tile row i that receives updates / of tile column j
__host__ __device__ inline int la_tile_m(const LaArgs &a, int i) { int64_t m = a.rows - (int64_t)NB * i; return (int)(m < 0 ? 0 : (m > NB ? NB : m)); }
__host__ __device__ inline int la_tile_n(const LaArgs &a, int j) { int64_t n = a.ncols - (int64_t)NB * j; return (int)(n < 0 ? 0 : (n > NB ? NB : n)); }
// 32 x 32 sub-tiles of an m x n tile: all of them / the ones on or above the diagonal (columns first)
__host__ __device__ inline int la_sub_rect(int m, int n) { return ((m + 31) >> 5) * ((n + 31) >> 5); }
__host__ __device__ inline int la_sub_diag(int m, int n)
{
	const int ti = (m + 31) >> 5, tj = (n + 31) >> 5;
	int c = 0;
	for(int j = 0; j < tj; ++ j)
		c += ((j < ti - 1) ? j : ti - 1) + 1;
	return ti > 0 ? c : 0;
}

constexpr int LA_THREADS = 1024;
constexpr int LA_LDS_DOUBLES = (NB + 16) * FS_STRIDE + 8; // the inverse (128 columns) + one 16-column slab; status words at the end
static_assert(LA_LDS_DOUBLES >= POTRF_LDS_DOUBLES && LA_LDS_DOUBLES >= (32 + 32) * FS_STRIDE, "chain kernel LDS");
static_assert(LA_LDS_DOUBLES * 8 <= 160 * 1024, "chain kernel LDS exceeds a CU");

__device__ __forceinline__ int la_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wait until cnt[i0] >= t0 and cnt[i1] >= t1 (an index < 0: no condition). ONE lane polls (relaxed, s_sleep), runs the
// agent-scope acquire and waits for it; the barrier holds the other waves' loads behind it. `phase` alternates the status
// word so that a second wait cannot overwrite a status some wave has not read yet. Returns false after an abort / timeout.
__device__ __forceinline__ bool la_wait(const LaArgs &a, int i0, int t0, int i1, int t1, double *sm, int &phase)
{
	int *okw = (int*)(sm + LA_LDS_DOUBLES - 2) + (phase & 1) * 2;
	phase ^= 1;
	if(threadIdx.x == 0) {
		int ok = 1;
		const long long tb = wall_clock64();
		for(int it = 0;; ++ it) {
			const bool r0 = i0 < 0 || la_ld(a.cnt + i0) >= t0;
			const bool r1 = i1 < 0 || la_ld(a.cnt + i1) >= t1;
			if(r0 && r1)
				break;
			if((it & 15) == 15) {
				if(la_ld(a.abort) != 0) {
					ok = 0;
					break;
				}
				if(wall_clock64() - tb > a.timeout_ticks) {
					__hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					ok = 0;
					break;
				}
			}
			__builtin_amdgcn_s_sleep(1);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		*okw = ok;
	}
	__syncthreads();
	return *okw != 0;
}

// every storing wave has drained its (write-through) stores, then ONE lane bumps the counter
__device__ __forceinline__ void la_publish(const LaArgs &a, int idx, int inc = 1)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if(threadIdx.x == 0)
		__hip_atomic_fetch_add(a.cnt + idx, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- row-panel slabs: X = Tinv^T Y in place, the inverse staged ONCE per step --------------------------------------
// As: the inverse, element (k, m) at As[m * FS_STRIDE + k] -- only k < 16 (m / 16 + 1) is ever read (upper triangular:
// Tinv(k, m) = 0 for k > m), so only that part is fetched
__device__ __forceinline__ void la_stage_tinv(const double *__restrict__ tinv, double *As)
{
	constexpr int KP = NB / 2; // 16-byte pieces per column
	double2 v[(NB * KP) / LA_THREADS];
#pragma unroll
	for(int i = 0; i < (NB * KP) / LA_THREADS; ++ i) {
		const int p = threadIdx.x + i * LA_THREADS, m = p / KP, q = p % KP;
		v[i] = (2 * q < 16 * (m / 16 + 1)) ? *(const double2*)(tinv + (size_t)m * NB + 2 * q) : make_double2(0, 0);
	}
#pragma unroll
	for(int i = 0; i < (NB * KP) / LA_THREADS; ++ i) {
		const int p = threadIdx.x + i * LA_THREADS, m = p / KP, q = p % KP;
		if(2 * q < 16 * (m / 16 + 1))
			*(double2*)(&As[m * FS_STRIDE + 2 * q]) = v[i];
	}
}

// the 16-byte piece of a slab this thread stages (1024 threads = 16 columns x 64 pieces); columns beyond nvalid are clamped
__device__ __forceinline__ double2 la_slab_fetch(const double *Y, int64_t ld, int nvalid)
{
	int col = threadIdx.x >> 6;
	const int q = threadIdx.x & 63;
	if(col > nvalid - 1)
		col = nvalid - 1;
	return *(const double2*)(Y + (int64_t)col * ld + 2 * q);
}

// one slab (128 x nvalid <= 16 columns at Y, leading dimension ld), `y` = this thread's staged piece of it.
// Row tile t (16 rows) only sees k < 16 (t + 1); waves t and t + 8 split that range in halves, waves 8..15 hand their
// partial tile over through LDS (the slab's own image, dead by then).
__device__ __forceinline__ void la_trsm_slab(const double *As, double *Bs, double *Y, int64_t ld, int nvalid, const double2 y)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
	*(double2*)(&Bs[(tid >> 6) * FS_STRIDE + 2 * (tid & 63)]) = y;
	__syncthreads();
	const int t = wave & 7, half = wave >> 3;
	const int kb = half * 8 * (t + 1), ke = kb + 8 * (t + 1);
	v4f64 acc = (v4f64){0, 0, 0, 0};
	for(int k4 = kb; k4 < ke; k4 += 4) {
		const double fa = As[(t * 16 + l15) * FS_STRIDE + k4 + l4];
		const double fb = Bs[l15 * FS_STRIDE + k4 + l4];
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb, fa, acc, 0, 0, 0);
	}
	__syncthreads(); // every wave is done with the slab image
	if(half) {
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			Bs[t * 256 + (l4 + 4 * r) * 16 + l15] = acc[r];
	}
	__syncthreads();
	if(!half) {
#pragma unroll
		for(int r = 0; r < 4; ++ r) {
			const int n = l4 + 4 * r;
			const double v = acc[r] + Bs[t * 256 + n * 16 + l15];
			if(n < nvalid)
				__hip_atomic_store(&Y[(t * 16 + l15) + (int64_t)n * ld], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// sub-tile t of an m x n tile (32 x 32 pieces, columns first; `diag`: only the pieces on or above the diagonal)
__device__ __forceinline__ void la_subtile_of(int t, int m, int n, bool diag, int &si, int &sj)
{
	const int ti = (m + 31) >> 5;
	if(!diag) {
		sj = t / ti;
		si = t % ti;
		return;
	}
	int j = 0, rem = t;
	for(;; ++ j) {
		const int cnt = ((j < ti - 1) ? j : ti - 1) + 1;
		if(rem < cnt)
			break;
		rem -= cnt;
	}
	sj = j;
	si = rem;
}

#define LA_TRACE(slot) do { if(a.trace && threadIdx.x == 0) a.trace[(slot)] = wall_clock64(); } while(0)

#ifndef LA_POTRF_STORE
#define LA_POTRF_STORE 0 // how potrf publishes R and the inverse: 0 plain stores + ONE agent-scope release, 1 write-through stores, 2 write-through without the zeros
#endif

// ---- workgroup 0: the diagonal blocks ------------------------------------------------------------------------------
// (a function of its own, not inlined: inside the loop over k the register allocator spilled 72 VGPRs of the
// factorization's 108 to scratch; as a callee the body keeps the allocation it has as a kernel)
__device__ __attribute__((noinline)) void la_potrf_step(double *Ablk, int64_t ld, int n_valid, int has_rhs, double *tinv, int *info,
	int64_t k0, double *sm)
{
	potrf_diag_body<false, LA_POTRF_STORE>(Ablk, ld, n_valid, has_rhs, tinv, info, k0, sm);
}

__device__ __forceinline__ void la_potrf_role(const LaArgs &a, double *sm)
{
	int phase = 0;
	for(int k = 0; k < a.nsteps; ++ k) {
		const int64_t k0 = (int64_t)NB * k;
		if(k > 0) {
			const int tgt = la_sub_diag(la_tile_m(a, k), la_tile_n(a, k));
			if(!la_wait(a, la_diag_cnt(a, k), tgt, -1, 0, sm, phase))
				return;
		}
		LA_TRACE(8 * k + 0);
		int64_t nv = a.n_id - k0;
		nv = nv < 0 ? 0 : (nv > NB ? NB : nv);
		la_potrf_step(a.A + k0 + k0 * a.ld, a.ld, (int)nv, (a.has_rhs && nv < NB) ? 1 : 0,
			a.tinv_all + (size_t)k * NB * NB, a.info, k0, sm);
		LA_TRACE(8 * k + 2);
		if(LA_POTRF_STORE == 0) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if(threadIdx.x == 0) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_fetch_add(a.cnt + la_potrf_done(a, k), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		} else
			la_publish(a, la_potrf_done(a, k));
		LA_TRACE(8 * k + 1);
	}
}

// ---- G1: the 2 x 2 window right of / below the diagonal block ---------------------------------------------------------
// workgroup g of the group: 0..7 take the slabs of b1, all ten a sub-tile of c1 each, 0..7 the slabs of b2 (the inverse
// staged again: c1 used the LDS), then the sub-tiles of c2 go round all of them. Nothing behind c1 is on the critical
// path: it has the duration of potrf(k + 1) to finish.
constexpr int LA_G1 = 10;

__device__ __forceinline__ void la_window_role(const LaArgs &a, const int g, double *sm)
{
	int phase = 0;
	double *As = sm, *Bs = sm + NB * FS_STRIDE;
	for(int k = 0; k < a.nsteps; ++ k) {
		const int64_t k0 = (int64_t)NB * k, c1 = k0 + NB, c2 = c1 + NB;
		if(c1 >= a.ncols)
			break;
		const double *tinv = a.tinv_all + (size_t)k * NB * NB;
		const int ns1 = la_nslabs(a, k + 1), ns2 = la_nslabs(a, k + 2);
		const int m1 = la_tile_m(a, k + 1), m2 = la_tile_m(a, k + 2), n1 = la_tile_n(a, k + 1), n2 = la_tile_n(a, k + 2);
		// ---- b1: slab g of R(k, k+1)
		if(g < 8 && g < ns1) {
			double *Y = a.A + k0 + (c1 + 16 * g) * a.ld;
			const int nvalid = (int)((a.ncols - (c1 + 16 * g) < 16) ? (a.ncols - (c1 + 16 * g)) : 16);
			// tile (k, k+1) is final once c2(k - 1) has updated it; the inverse comes with potrf(k)
			if(!la_wait(a, la_potrf_done(a, k), 1, k > 0 ? la_win_cnt(a, k - 1, 0) : -1, k > 0 ? la_sub_rect(la_tile_m(a, k), n1) : 0, sm, phase))
				return;
			if(g == 0) LA_TRACE(8 * k + 3);
			const double2 y = la_slab_fetch(Y, a.ld, nvalid);
			la_stage_tinv(tinv, As);
			la_trsm_slab(As, Bs, Y, a.ld, nvalid, y);
			la_publish(a, la_col_cnt(a, k, k + 1));
			if(g == 0) LA_TRACE(8 * k + 4);
		}
		// ---- c1: sub-tile g of S(k+1, k+1) -= R(k, k+1)^T R(k, k+1)
		if(m1 > 0) {
			const int nsub = la_sub_diag(m1, n1);
			if(g < nsub) {
				// the slabs of R(k, k+1); the tile's previous update came from c2(k - 1)
				if(!la_wait(a, la_col_cnt(a, k, k + 1), ns1, k > 0 ? la_win_cnt(a, k - 1, 1) : -1, k > 0 ? la_sub_diag(m1, n1) : 0, sm, phase))
					return;
				if(g == 0) LA_TRACE(8 * k + 5);
				const double *P = a.A + k0 + c1 * a.ld;
				for(int t = g; t < nsub; t += LA_G1) {
					int si, sj;
					la_subtile_of(t, m1, n1, true, si, sj);
					gemm_tn_staged_tile<32, 32, 16, 16, 0, 0, 1, LA_THREADS>((int64_t)si * 32, (int64_t)sj * 32, m1, n1, P, a.ld, P, a.ld,
						a.A + c1 + c1 * a.ld, a.ld, sm);
					la_publish(a, la_diag_cnt(a, k + 1));
				}
				if(g == 0) LA_TRACE(8 * k + 6);
			}
		}
		// ---- b2: slab g of R(k, k+2): the tile got its last update from bulk(k - 1) (first tile row)
		if(g < 8 && g < ns2) {
			double *Y = a.A + k0 + (c2 + 16 * g) * a.ld;
			const int nvalid = (int)((a.ncols - (c2 + 16 * g) < 16) ? (a.ncols - (c2 + 16 * g)) : 16);
			if(!la_wait(a, la_potrf_done(a, k), 1, k > 0 ? la_bt_cnt(a, k - 1, 0, k + 2) : -1, 4, sm, phase))
				return;
			const double2 y = la_slab_fetch(Y, a.ld, nvalid);
			la_stage_tinv(tinv, As);
			la_trsm_slab(As, Bs, Y, a.ld, nvalid, y);
			la_publish(a, la_col_cnt(a, k, k + 2));
		}
		// ---- c2: S(k+1, k+2) -= R(k, k+1)^T R(k, k+2) and S(k+2, k+2) -= R(k, k+2)^T R(k, k+2)
		if(ns2 > 0 && m1 > 0) {
			const int nsa = la_sub_rect(m1, n2), nsb = (m2 > 0) ? la_sub_diag(m2, n2) : 0;
			bool waited_a = false, waited_b = false;
			for(int t = g; t < nsa + nsb; t += LA_G1) {
				if(t < nsa) {
					if(!waited_a) {
						// both panels; the tile's previous update came from bulk(k - 1) (its second tile row)
						if(!la_wait(a, la_col_cnt(a, k, k + 1), ns1, la_col_cnt(a, k, k + 2), ns2, sm, phase))
							return;
						if(k > 0 && !la_wait(a, la_bt_cnt(a, k - 1, 1, k + 2), 4, -1, 0, sm, phase))
							return;
						waited_a = true;
					}
					int si, sj;
					la_subtile_of(t, m1, n2, false, si, sj);
					gemm_tn_staged_tile<32, 32, 16, 16, 0, 0, 1, LA_THREADS>((int64_t)si * 32, (int64_t)sj * 32, m1, n2,
						a.A + k0 + c1 * a.ld, a.ld, a.A + k0 + c2 * a.ld, a.ld, a.A + c1 + c2 * a.ld, a.ld, sm);
					la_publish(a, la_win_cnt(a, k, 0));
				} else {
					if(!waited_b) {
						if(!la_wait(a, la_col_cnt(a, k, k + 2), ns2, k > 0 ? la_bt_cnt(a, k - 1, 2, k + 2) : -1, 4, sm, phase))
							return;
						waited_b = true;
					}
					int si, sj;
					la_subtile_of(t - nsa, m2, n2, true, si, sj);
					const double *P = a.A + k0 + c2 * a.ld;
					gemm_tn_staged_tile<32, 32, 16, 16, 0, 0, 1, LA_THREADS>((int64_t)si * 32, (int64_t)sj * 32, m2, n2, P, a.ld, P, a.ld,
						a.A + c2 + c2 * a.ld, a.ld, sm);
					la_publish(a, la_win_cnt(a, k, 1));
				}
			}
		}
	}
}

// ---- G2: the rest of row panel k -------------------------------------------------------------------------------------
// A workgroup takes whole tile columns: ONE wait, the whole 128 x 128 tile fetched into registers at once (8 pieces per
// thread, one memory round trip), eight slab products out of LDS, ONE publish. The columns of a panel are handed out
// through a counter (first come, first served, left to right), so the group makes progress with however many of its
// workgroups are resident.
__device__ __forceinline__ void la_panel_role(const LaArgs &a, double *sm)
{
	int phase = 0;
	double *As = sm, *Bs = sm + NB * FS_STRIDE;
	int *jw = (int*)(sm + LA_LDS_DOUBLES - 4); // the column this workgroup drew
	for(int k = 0; k < a.nsteps; ++ k) {
		const int64_t k0 = (int64_t)NB * k;
		if((int64_t)NB * (k + 3) >= a.ncols)
			break;
		bool staged = false;
		for(;;) {
			if(threadIdx.x == 0)
				*jw = k + 3 + __hip_atomic_fetch_add(a.cnt + la_pan_next(a, k), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__syncthreads();
			const int j = *jw;
			__syncthreads(); // (the word is rewritten at the top of the loop)
			if(j >= a.nt)
				break;
			const int ns = la_nslabs(a, j);
			// the inverse; tile (k, j) final: its last update came from bulk(k - 1) (first tile row)
			if(!la_wait(a, la_potrf_done(a, k), 1, k > 0 ? la_bt_cnt(a, k - 1, 0, j) : -1, 4, sm, phase))
				return;
			const int64_t cj = (int64_t)NB * j;
			double2 y[8];
#pragma unroll
			for(int s = 0; s < 8; ++ s) {
				const int64_t c = cj + 16 * s;
				const int nvalid = (int)((a.ncols - c < 16) ? (a.ncols - c) : 16);
				y[s] = (s < ns) ? la_slab_fetch(a.A + k0 + c * a.ld, a.ld, nvalid) : make_double2(0, 0);
			}
			if(!staged) {
				la_stage_tinv(a.tinv_all + (size_t)k * NB * NB, As);
				staged = true;
			}
#pragma unroll
			for(int s = 0; s < 8; ++ s) {
				if(s < ns) {
					const int64_t c = cj + 16 * s;
					const int nvalid = (int)((a.ncols - c < 16) ? (a.ncols - c) : 16);
					la_trsm_slab(As, Bs, a.A + k0 + c * a.ld, a.ld, nvalid, y[s]);
					__syncthreads(); // the slab image (and the partial tiles in it) is rewritten by the next slab
				}
			}
			la_publish(a, la_col_cnt(a, k, j), ns);
			if(a.trace && threadIdx.x == 0)
				atomicMax((unsigned long long*)a.trace + 8 * k + 7, (unsigned long long)wall_clock64());
		}
	}
}

__global__ __launch_bounds__(LA_THREADS)
void la_chain_kernel(const LaArgs a)
{
	extern __shared__ __attribute__((aligned(16))) double la_sm[];
	const int b = (int)blockIdx.x;
#ifndef LA_ROLES
#define LA_ROLES 7
#endif
	if(b == 0) {
		if(LA_ROLES & 1) la_potrf_role(a, la_sm);
	} else if(b <= LA_G1) {
		if(LA_ROLES & 2) la_window_role(a, b - 1, la_sm);
	} else {
		if(LA_ROLES & 4) la_panel_role(a, la_sm);
	}
}

constexpr int LA_BULK_LDS_DOUBLES = T444_LDS_DOUBLES + 2; // the 4x4x4 tile's two slab buffers (>= the 16x16x4 tile and the quarter path) + a status word

// ---- bulk update of step k ------------------------------------------------------------------------------------------
// region tiles (ri, rj) = matrix tiles (k+1+ri, k+1+rj), ri <= rj; the window (0,0), (0,1), (1,1) belongs to the chain.
// Block order: [priority tiles: region row 0 from column 2 on (the next row panel), then (1, 2) and (2, 2) (the next
//               window), as 64 x 64 quarters stored write-through, each signalling bt_cnt]
//              [the other tiles, columns first: the first n128 whole, the rest as quarters (fine-grained tail)]
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(SPP_MIXED_WAVES, SPP_MIXED_WAVES)))
void la_bulk_kernel(const LaArgs a, const int k, const int nr, const int nc, const int npri, const int64_t n128)
{
	extern __shared__ __attribute__((aligned(16))) double gemm_lds[];
	int &ok_s = *(int*)(gemm_lds + LA_BULK_LDS_DOUBLES - 2); // (dynamic LDS: a static variable would shift the 16-byte alignment of the images)
	const int64_t b = blockIdx.x;
	int ri, rj, quarter = -1, pri_row = -1;
	if(b < 4 * (int64_t)npri) {
		const int p = (int)(b >> 2);
		quarter = (int)(b & 3);
		if(p < nc - 2) {
			ri = 0;
			rj = 2 + p;
		} else {
			ri = 1 + (p - (nc - 2)); // (1, 2), then (2, 2)
			rj = 2;
		}
		pri_row = ri;
	} else {
		// the upper tiles of the region without its first tile row and column; three of them are not ours: (1,1) is the
		// chain's, (1,2) and (2,2) are priority tiles
		const int64_t bb = b - 4 * (int64_t)npri;
		int ti, tj;
		if(bb < n128)
			upper_tile_of(bb, nr - 1, ti, tj);
		else {
			const int64_t q = bb - n128;
			upper_tile_of(n128 + (q >> 2), nr - 1, ti, tj);
			quarter = (int)(q & 3);
		}
		ri = ti + 1;
		rj = tj + 1;
		if(rj <= 2)
			return;
	}
	const int gi = k + 1 + ri, gj = k + 1 + rj;
	// the two row-panel pieces this tile multiplies: every 16-column slab of R(k, gi) and R(k, gj) solved
	if(threadIdx.x == 0) {
		int ok = 1;
		const int i0 = la_col_cnt(a, k, gi), t0 = la_nslabs(a, gi), i1 = la_col_cnt(a, k, gj), t1 = la_nslabs(a, gj);
		const long long tb = wall_clock64();
		for(int it = 0; la_ld(a.cnt + i0) < t0 || la_ld(a.cnt + i1) < t1; ++ it) {
			if((it & 15) == 15) {
				if(la_ld(a.abort) != 0) {
					ok = 0;
					break;
				}
				if(wall_clock64() - tb > a.timeout_ticks) {
					__hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					ok = 0;
					break;
				}
			}
			__builtin_amdgcn_s_sleep(2);
		}
		if(a.bulk_acquire) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		ok_s = ok;
	}
	__syncthreads();
	if(!ok_s)
		return;
	long long t_go = 0;
	if(a.trace && threadIdx.x == 0) {
		t_go = wall_clock64();
		unsigned long long *tr = (unsigned long long*)a.trace + 8 * a.nsteps + 8 + 4 * k;
		atomicMin(tr + 0, (unsigned long long)t_go);
	}
	const int64_t c1 = (int64_t)NB * (k + 1);
	const int64_t M = a.rows - c1, N = a.ncols - c1; // region extents
	const double *P = a.A + (int64_t)NB * k + c1 * a.ld; // row panel k, region columns
	double *C = a.A + c1 + c1 * a.ld;
	bool skip = false;
	if(quarter < 0) {
		if(a.use444 && (int64_t)(ri + 1) * 128 <= M && (int64_t)(rj + 1) * 128 <= N) // (edge tiles: the bounds-checked tile)
		{
			if(a.use444 == 2)
				gemm_tn_tile_444<0>((int64_t)ri * 128, (int64_t)rj * 128, NB, P, a.ld, P, a.ld, C, a.ld, gemm_lds);
			else
				gemm_tn_tile_dma<0>((int64_t)ri * 128, (int64_t)rj * 128, NB, P, a.ld, P, a.ld, C, a.ld, gemm_lds);
		} else
			gemm_tn_tile<128, 128, 32, 32, 0, 1, 16>((int64_t)ri * 128, (int64_t)rj * 128, M, N, NB, P, a.ld, P, a.ld, C, a.ld, gemm_lds);
	}
	else {
		const int64_t m0 = (int64_t)ri * 128 + (quarter & 1) * 64, n0 = (int64_t)rj * 128 + ((quarter >> 1) & 1) * 64;
		skip = m0 >= n0 + 64 || m0 >= M || n0 >= N; // quarter strictly below the diagonal or outside the matrix
		if(!skip) {
			if(pri_row >= 0)
				gemm_tn_tile<64, 64, 16, 16, 0, 1, 32, 1>(m0, n0, M, N, NB, P, a.ld, P, a.ld, C, a.ld, gemm_lds);
			else
				gemm_tn_tile<64, 64, 16, 16, 0, 1, 32>(m0, n0, M, N, NB, P, a.ld, P, a.ld, C, a.ld, gemm_lds);
		}
	}
	if(a.trace && threadIdx.x == 0) {
		unsigned long long *tr = (unsigned long long*)a.trace + 8 * a.nsteps + 8 + 4 * k;
		const long long t_end = wall_clock64();
		atomicMax(tr + 1, (unsigned long long)t_end);
		atomicAdd(tr + 2, (unsigned long long)(t_end - t_go));
		if(pri_row >= 0)
			atomicMax(tr + 3, (unsigned long long)t_end);
	}
	if(pri_row >= 0) {
		// the chain reads this tile one or two steps from now: write-through stores, every wave drains, ONE lane counts
		// (a release fence here -- buffer_wbl2 by 470 workgroups per launch -- wrote back the XCDs' whole L2 each time)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if(threadIdx.x == 0)
			__hip_atomic_fetch_add(a.cnt + la_bt_cnt(a, k, pri_row, gj), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

} // namespace spp
