// spp_dense_dev.h -- device-side building blocks of the dense factor that more than one translation unit uses:
// the fully staged MFMA tile product (row-panel solve / trailing update of the chain kernels) and the factorization of
// one 128 x 128 diagonal block in LDS. spp_dense.hip launches them as kernels of their own and inside its fused chain
// kernel; spp_sparse.hip runs them inside the dependency-driven frontal kernel (teams of workgroups on the big fronts).
#pragma once
#include "spp_internal.h"
#include "spp_tiles.h"

// The fp64 MFMA form used by the tile products is v_mfma_f64_16x16x4 (2048 flop). Measured on MI355X with the operand
// pattern of a real tile product (tools/mfma_rate2.hip: a register outer product, distinct operand registers per
// instruction) it issues once per ~75 cycles and SIMD = 67 TFLOP/s chip-wide; v_mfma_f64_4x4x4_4b (four independent
// 4 x 4 x 4 blocks, 512 flop; lane maps: A[blk][i][k] in lane 16 k + 4 blk + i, B[blk][k][j] in lane 16 k + 4 blk + j,
// D[blk][i][j] in lane 16 i + 4 blk + j) once per 16-17 cycles = 77 TFLOP/s -- 14 % apart, at four times the operand
// words per flop. The tile built around the small form (spp_dense_444.h) measures slower; it is kept there, selectable.

namespace spp {

// workgroup barrier that orders LDS accesses only (s_waitcnt lgkmcnt(0) + s_barrier): unlike __syncthreads()
// it does not wait for outstanding global stores
__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Fully staged variant for the latency-critical chain kernels (row-panel solve and tile-row update):
// K <= 128, both operand panels are loaded into LDS ONCE (all global loads in flight together, one
// barrier), then the MFMAs run without further synchronization. LDS row stride K_MAX + 2 doubles:
// conflict-free ds_read_b64 fragments (lanes l & 15 step 4 banks, lanes l >> 4 step 2 banks).
constexpr int FS_KMAX = 128, FS_STRIDE = FS_KMAX + 2;

// one BM x BN tile of C at (m0, n0) by the first (BM / WM) * (BN / WN) waves of the workgroup (the body shared by
// gemm_tn_staged_kernel and the fused update + diagonal-block kernel); fs_lds: (BM + BN) * FS_STRIDE doubles
// SC1: the C tile is stored with agent-scope atomic (write-through) stores -- for a tile that another workgroup of the
// SAME launch reads with agent-scope atomic loads after a counter hand-off (no release / acquire fences, see
// update_potrf_kernel)
// PART (= threads of the workgroup, or 0): the workgroup has MORE waves than the tile uses and keeps them (a persistent
// workgroup that goes on to other work): ALL of its threads stage the operands -- half the staging registers per
// thread, twice the loads in flight --, the waves beyond the first (BM / WM) * (BN / WN) skip the MFMAs
template <int BM, int BN, int WM, int WN, int MODE, int ATRI, int SC1 = 0, int PART = 0>
__device__ __forceinline__ void gemm_tn_staged_tile(const int64_t m0, const int64_t n0, int64_t M, int64_t N,
	const double *__restrict__ A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, double *fs_lds)
{
	constexpr bool a_upper_tri = ATRI != 0; // A(k, m) = 0 for k > m, BM covers all of A's columns (m0 == 0)
	constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
	constexpr int TA = WM / 16, TB = WN / 16;
	double *As = fs_lds, *Bs = fs_lds + BM * FS_STRIDE;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = (wave % NWM) * WM, wn = (wave / NWM) * WN;
	const int l15 = lane & 15, l4 = lane >> 4;
	// stage A (BM columns) and B (BN columns): piece p -> (column p / 64, piece p % 64); K == FS_KMAX.
	// All loads of a thread are issued before its first LDS store (one exposed memory round trip).
	constexpr int NTS = PART ? PART : NT; // threads that stage
	constexpr int KP = FS_KMAX / 2, PA = (BM * KP) / NTS, PB = (BN * KP) / NTS;
	static_assert((BM * KP) % NTS == 0 && (BN * KP) % NTS == 0, "staging must divide evenly");
	const bool live = !PART || tid < NT;
	{
		double2 va[PA], vb[PB];
#pragma unroll
		for(int i = 0; i < PA; ++ i) {
			const int p = tid + i * NTS, col = p / KP, q = p % KP;
			int64_t gc = m0 + col;
			if(gc > M - 1) gc = M - 1;
			// a_upper_tri: A(k, m) = 0 for k > m (the inverse of a diagonal block): the zero half is not fetched
			va[i] = (a_upper_tri && 2 * q > gc) ? make_double2(0, 0) : *(const double2*)(A + gc * lda + 2 * q);
		}
#pragma unroll
		for(int i = 0; i < PB; ++ i) {
			const int p = tid + i * NTS, col = p / KP, q = p % KP;
			int64_t gc = n0 + col;
			if(gc > N - 1) gc = N - 1;
			vb[i] = *(const double2*)(B + gc * ldb + 2 * q);
		}
#pragma unroll
		for(int i = 0; i < PA; ++ i) {
			const int p = tid + i * NTS, col = p / KP, q = p % KP;
			*(double2*)(&As[col * FS_STRIDE + 2 * q]) = va[i];
		}
#pragma unroll
		for(int i = 0; i < PB; ++ i) {
			const int p = tid + i * NTS, col = p / KP, q = p % KP;
			*(double2*)(&Bs[col * FS_STRIDE + 2 * q]) = vb[i];
		}
	}
	v4f64 acc[TB][TA];
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a) {
			acc[b][a] = (v4f64){0, 0, 0, 0};
			if(MODE == 0 && live) {
				const int64_t m = m0 + wm + a * 16 + l15;
#pragma unroll
				for(int r = 0; r < 4; ++ r) {
					const int64_t n = n0 + wn + b * 16 + l4 + 4 * r;
					if(m < M && n < N)
						acc[b][a][r] = -C[m + n * ldc];
				}
			}
		}
	__syncthreads();
	if(!live)
		return;
	// rows wm .. wm + WM - 1 of an upper triangular A^T only see k < wm + WM (wave-uniform bound)
	const int kend = (a_upper_tri && wm + WM < FS_KMAX) ? wm + WM : FS_KMAX; // compile-time FS_KMAX unless ATRI
#pragma unroll 4
	for(int k4 = 0; k4 < kend; k4 += 4) {
		double fa[TA], fb[TB];
#pragma unroll
		for(int a = 0; a < TA; ++ a)
			fa[a] = As[(wm + a * 16 + l15) * FS_STRIDE + k4 + l4];
#pragma unroll
		for(int b = 0; b < TB; ++ b)
			fb[b] = Bs[(wn + b * 16 + l15) * FS_STRIDE + k4 + l4];
#pragma unroll
		for(int b = 0; b < TB; ++ b)
#pragma unroll
			for(int a = 0; a < TA; ++ a)
				acc[b][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[b][a], 0, 0, 0);
	}
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a) {
			const int64_t m = m0 + wm + a * 16 + l15;
#pragma unroll
			for(int r = 0; r < 4; ++ r) {
				const int64_t n = n0 + wn + b * 16 + l4 + 4 * r;
				if(m < M && n < N) {
					const double val = (MODE == 0) ? -acc[b][a][r] : acc[b][a][r];
					if(SC1)
						__hip_atomic_store(&C[m + n * ldc], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					else
						C[m + n * ldc] = val;
				}
			}
		}
}

// Row-panel solve of one 16-column slab, X = R_kk^-T Y in place, with the diagonal block's inverse in the HALF form
// potrf_diag_body<.., .., 1> leaves: tinv = [T0  R01; 0  T1] (T0, T1 the inverses of R_kk's two 64 x 64 diagonal
// blocks, R01 its off-diagonal block):
//     X0 = T0^T Y0 ;  W = Y1 - R01^T X0 ;  X1 = T1^T W
// One 16 x 16 MFMA tile of X per wave (waves 0..3: X0, 4..7: X1), the slab of Y and the 128 x 128 operand fully staged
// in LDS as in gemm_tn_staged_tile (same layout and size: (128 + 16) * FS_STRIDE doubles); the two halves hand over
// through the staged slab itself (three workgroup barriers). 36 tile products per slab -- the triangles are used,
// against 64 of the dense product with the full inverse -- in a dependent chain of at most 4 + 4 + 4 tiles.
// THREADS: threads of the calling workgroup (>= 512); all of them stage, the first 8 waves compute.
template <int THREADS, int SC1 = 0>
__device__ __forceinline__ void panel_solve_slab(const int64_t n0, const int64_t N, const double *__restrict__ tinv,
	double *Y, const int64_t ld, double *fs_lds)
{
	static_assert(THREADS >= 512 && THREADS % 64 == 0, "panel_solve_slab: eight waves compute");
	constexpr int KP = FS_KMAX / 2, PA = (FS_KMAX * KP) / THREADS, PB = (16 * KP + THREADS - 1) / THREADS;
	static_assert((FS_KMAX * KP) % THREADS == 0, "staging must divide evenly");
	double *As = fs_lds, *Bs = fs_lds + FS_KMAX * FS_STRIDE;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int l15 = lane & 15, l4 = lane >> 4, wm = wave * 16;
	{
		double2 va[PA], vb[PB];
#pragma unroll
		for(int i = 0; i < PA; ++ i) {
			const int p = tid + i * THREADS, m = p / KP, q = p % KP;
			// column m of the operand: rows 0 .. m (to the end of its 16-row tile: zeros there, stored by the factorization)
			va[i] = (2 * q < 16 * (m / 16 + 1)) ? *(const double2*)(tinv + (size_t)m * FS_KMAX + 2 * q) : make_double2(0, 0);
		}
#pragma unroll
		for(int i = 0; i < PB; ++ i) {
			const int p = tid + i * THREADS, col = p / KP, q = p % KP;
			int64_t gc = n0 + col;
			if(gc > N - 1) gc = N - 1;
			if(p < 16 * KP)
				vb[i] = *(const double2*)(Y + gc * ld + 2 * q);
		}
#pragma unroll
		for(int i = 0; i < PA; ++ i) {
			const int p = tid + i * THREADS, m = p / KP, q = p % KP;
			*(double2*)(&As[m * FS_STRIDE + 2 * q]) = va[i];
		}
#pragma unroll
		for(int i = 0; i < PB; ++ i) {
			const int p = tid + i * THREADS, col = p / KP, q = p % KP;
			if(p < 16 * KP)
				*(double2*)(&Bs[col * FS_STRIDE + 2 * q]) = vb[i];
		}
	}
	__syncthreads();
	const double *am = As + (wm + l15) * FS_STRIDE + l4, *bn = Bs + l15 * FS_STRIDE + l4;
	auto chain = [&](const int kb, const int ke) { // sum over k = kb .. ke - 1 of A(k, wm + .)^T B(k, .)
		v4f64 acc = (v4f64){0, 0, 0, 0};
#pragma unroll 4
		for(int k4 = kb; k4 < ke; k4 += 4)
			acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bn[k4], am[k4], acc, 0, 0, 0);
		return acc;
	};
	auto store_out = [&](const v4f64 x) {
#pragma unroll
		for(int r = 0; r < 4; ++ r) {
			const int64_t n = n0 + l4 + 4 * r;
			if(n < N) {
				if(SC1)
					__hip_atomic_store(&Y[wm + l15 + n * ld], x[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				else
					Y[wm + l15 + n * ld] = x[r];
			}
		}
	};
	double *own = Bs + l4 * FS_STRIDE + wm + l15; // this lane's elements of the slab: (row wm + l15, column l4 + 4 r)
	v4f64 x = (v4f64){0, 0, 0, 0};
	if(wave < 4)
		x = chain(0, wm + 16); // X0 = T0^T Y0
	lds_barrier(); // Y0 has been read
	if(wave < 4) {
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			own[4 * r * FS_STRIDE] = x[r];
	}
	lds_barrier();
	if(wave < 4)
		store_out(x);
	else if(wave < 8) {
		const v4f64 u = chain(0, 64); // R01^T X0
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			own[4 * r * FS_STRIDE] -= u[r]; // W, in place (each wave only reads its own 16 rows of Y1 here)
	}
	lds_barrier();
	if(wave >= 4 && wave < 8)
		store_out(chain(64, wm + 16)); // X1 = T1^T W
}

// --------------------------------------------------------------------------------------------------
// potrf of one 128 x 128 diagonal block, entirely in LDS, blocked by 16 with MFMA f64 updates.
//   in : T = upper triangle of the block (global, column-major, ld); padding rows/cols (>= n_valid)
//        are exact identity; if has_rhs, column n_valid holds a right-hand side (rows < n_valid)
//   out: upper triangle <- R_kk ; rhs column <- R_kk^-T rhs ;
//        tinv (128 x 128 column-major, dense upper triangular) <- R_kk^-1
// Per 16-wide panel J (16 waves):
//   B  row panel: X = Dinv^T Y for the tiles right of the diagonal (R part) and left of it (G part:
//      the rows of (R^-1)^T accumulated in the unused LOWER triangle of the block); one tile per wave
//   C  trailing update T[I,K] -= P_I^T P_K (R part), G[I,Cb] -= P_I^T G[J,Cb] (G part): waves 1..15
//      take the tiles two at a time (independent MFMA chains); wave 0 updates the NEXT diagonal tile
//      first and then
//   A  factors and inverts that 16 x 16 tile in registers (square-root-free elimination, 5 double
//      shuffles per pivot) while the other waves finish C -- the serial part hides under the update.
// 2 workgroup barriers per panel. Dinv / G_JJ scratch tiles are double buffered (panel parity).
// info[0] = first failing global pivot index + 1 (non-positive pivot, Eigen's LLT test).
// --------------------------------------------------------------------------------------------------
#ifdef SPP_POTRF_TRACE
__device__ long long spp_potrf_trace[64]; // cycle stamps of thread 0 / thread 64 (tools/potrf_trace.hip)
#define SPP_STAMP(slot, who) do { if(tid == (who)) spp_potrf_trace[slot] = (long long)__builtin_readcyclecounter(); } while(0)
#else
#define SPP_STAMP(slot, who) do { } while(0)
#endif

constexpr int NB = DENSE_NB;
#ifndef SPP_POTRF_TS
#define SPP_POTRF_TS (NB + 1)
#endif
constexpr int TS = SPP_POTRF_TS;   // LDS column stride of the block image: element (r, c) at r + c * TS
constexpr int POTRF_THREADS = 1024;
constexpr int POTRF_LDS_DOUBLES = NB * TS + 4 * 16 * PT + 2 * NB + 8;
constexpr int POTRF_LDS_DOUBLES_INV2 = NB * TS + (2 + NB / 16) * 16 * PT + 2 * NB + 8; // HALF = 2: a G_JJ slot per panel
static_assert(POTRF_LDS_DOUBLES_INV2 * 8 <= 160 * 1024, "the diagonal-block factorization's LDS exceeds a CU");

// HALF = 2 streams the factorization out: after panel J the rows 16 J .. 16 J + 15 of R (stored write-through) and the
// inverse of their diagonal tile (dbuf + 256 J, Dinv[k][i] at k + 16 i) are complete in memory and *flag = base + J + 1
// LDS pointers of potrf_diag_body<.., .., 2> (for a caller that factors the first tile itself: first_done)
__device__ __forceinline__ double *potrf_lds2_dv(double *sm) { return sm + DENSE_NB * SPP_POTRF_TS; }
__device__ __forceinline__ double *potrf_lds2_gd0(double *sm) { return potrf_lds2_dv(sm) + 2 * 16 * PT; }
__device__ __forceinline__ double *potrf_lds2_dinv(double *sm) { return potrf_lds2_dv(sm) + (2 + DENSE_NB / 16) * 16 * PT; }
__device__ __forceinline__ int *potrf_lds2_fail(double *sm) { return (int*)(potrf_lds2_dinv(sm) + 2 * DENSE_NB); }

struct PotrfPub {
	int *flag = nullptr;
	int base = 0;
	double *dbuf = nullptr;
	int *abort = nullptr; // raised when the block is not positive definite (the consumers of the stream give up)
};

// COH: the block was written by other workgroups of the same launch with write-through stores: read it with
// agent-scope atomic (sc1) loads -- they bypass this CU's L1, which a plain load could be served from stale
template <bool COH>
__device__ __forceinline__ double2 ld_blk2(const double *p)
{
	if(COH) {
		double2 v;
		v.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		v.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return v;
	}
	return *(const double2*)p;
}

// WT: R and the inverse are stored write-through (agent-scope atomic = sc1 stores): the block is handed to other workgroups
// of the same launch behind a drained flag, without a release fence that would write back the whole L2 (lookahead chain)
template <int WT>
__device__ __forceinline__ void st_blk(double *p, const double v)
{
	if(WT)
		__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else
		*p = v;
}

// WT = 2: as 1, and the zeros below the diagonal of the inverse are NOT stored (the buffer is zeroed when it is allocated
// and nothing else is ever written there): a third fewer stores behind the factorization
// HALF = 1: `tinv` receives the inverses of the two 64 x 64 diagonal blocks of R_kk and, between them (rows < 64, columns
// >= 64), R_kk's own off-diagonal block R01 -- not the 64 x 64 block -T0 R01 T1 of the full inverse. The rows of the
// inverse are accumulated inside the panel loop (G part); of its 84 tile updates 64 belong to that block, and the first
// panels -- where the R part alone outlasts wave 0's chain -- carry most of them. A row-panel solve with this form runs
// in two dependent halves of the same total size: X0 = T0^T Y0 ; X1 = T1^T (Y1 - R01^T X0)  (panel_solve_slab).
// HALF = 2 (the streamed tail of the dense factor, spp_dense_tail.h): NO rows of the inverse inside the panel loop -- the
// loop is the factorization alone, 46 k cycles without its prologue --, every finished row tile is published (PotrfPub),
// and the whole inverse is formed AFTER the loop, off the critical chain, by recursive doubling over the 16 x 16 tiles.
// PRELOADED: the block's upper triangle is already in the LDS image (the workgroup accumulated it there).
template <bool COH = false, int WT = 0, int HALF = 0, bool PRELOADED = false>
__device__ __forceinline__ void potrf_diag_body(double *__restrict__ Ablk, int64_t ld, int n_valid, int has_rhs,
	double *__restrict__ tinv, int *__restrict__ info, int64_t k0, double *sm, const PotrfPub pub = PotrfPub(), const int first_done = 0)
{
	// first_done (PRELOADED only): the caller has factored the first 16 x 16 tile in place already (diag_tile_factor on the
	// image, with the LDS pointers of potrf_lds2_*, *fail initialized) -- the streamed launch does that under the last
	// updates of the tile
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	SPP_STAMP(0, 0);
	constexpr int GD_SLOTS = (HALF == 2) ? NB / 16 : 2;
	double *T = sm;                  // NB x NB image, stride TS
	double *DvB = T + NB * TS;       // 2 x Dinv[k][i] at Dv[k + i * PT] (upper triangular, zeros below)
	double *GdB = DvB + 2 * 16 * PT; // G_JJ[r][c] = Dinv[c][r] at Gd[r + c * PT] (lower triangular incl. diagonal): two slots by panel parity, or (HALF = 2) one per panel
	auto gd_slot = [&](int J) { return GdB + ((HALF == 2) ? J : (J & 1)) * 16 * PT; };
	double *dinv = GdB + GD_SLOTS * 16 * PT; // 1 / R[j][j]
	double *yv = dinv + NB;          // carried right-hand side
	int *fail = (int*)(yv + NB);
	const int l15 = lane & 15, l4 = lane >> 4;
	constexpr int NW = POTRF_THREADS / 64;
	const int rhs_col = (has_rhs && n_valid < NB) ? n_valid : -1;
	// Prologue. The first diagonal tile only needs its own 16 x 16 entries: wave 0 fetches them and starts
	// the elimination (5 500 cycles) while the other 15 waves stream in the rest of the block (6 500 cycles).
	// Not possible when the carried right-hand side sits inside that tile (n_valid < 16): plain order then.
	const bool fast0 = !PRELOADED && !(rhs_col >= 0 && rhs_col < 16);
	if(tid == 0) {
		if(!first_done)
			fail[0] = 0;
		fail[2] = 0; // (HALF = 2: count of the waves that have their part of a streamed row tile in memory)
	}
	if(fast0) {
		if(wave == 0) {
			double v[4];
#pragma unroll
			for(int t = 0; t < 4; ++ t)
				v[t] = COH ? __hip_atomic_load(&Ablk[l15 + (int64_t)(l4 + 4 * t) * ld], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
				           : Ablk[l15 + (int64_t)(l4 + 4 * t) * ld];
#pragma unroll
			for(int t = 0; t < 4; ++ t)
				T[l15 + (l4 + 4 * t) * TS] = v[t];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			diag_tile_factor<TS>(T, DvB, gd_slot(0), dinv, 0, lane, fail, info, k0);
		} else {
			// 16-byte pieces of the block without its first tile, spread over the 960 threads of waves 1..15;
			// all of a thread's loads are in flight before its first LDS store
			constexpr int NP = NB * NB / 2, NTH = POTRF_THREADS - 64, NIT = (NP + NTH - 1) / NTH;
			double2 v[NIT];
#pragma unroll
			for(int t = 0; t < NIT; ++ t) {
				const int e = (tid - 64) + t * NTH, r = (e & (NB / 2 - 1)) * 2, c = e >> 6;
				// only the upper triangle is an input (16 x 16 tiles on or above the diagonal): the tiles below it
				// are first ASSIGNED by the G part of the update, never read before -- 44 % of the block not fetched
				if(e < NP && !(r < 16 && c < 16) && (r >> 4) <= (c >> 4))
					v[t] = ld_blk2<COH>(Ablk + r + (int64_t)c * ld);
			}
#pragma unroll
			for(int t = 0; t < NIT; ++ t) {
				const int e = (tid - 64) + t * NTH, r = (e & (NB / 2 - 1)) * 2, c = e >> 6;
				if(e < NP && !(r < 16 && c < 16) && (r >> 4) <= (c >> 4)) {
					T[r + c * TS] = v[t].x;
					T[r + 1 + c * TS] = v[t].y;
				}
			}
		}
		__syncthreads();
		SPP_STAMP(1, 0);
		if(tid < NB) {
			yv[tid] = (rhs_col >= 0 && tid < n_valid) ? T[tid + rhs_col * TS] : 0.0;
			if(tid >= 16)
				dinv[tid] = 1.0; // entries 0..15 were set by the elimination of the first tile
		}
		__syncthreads();
		if(rhs_col >= 0 && tid < NB)
			T[tid + rhs_col * TS] = (tid == rhs_col) ? 1.0 : 0.0; // the rhs column becomes plain padding
		__syncthreads();
	} else {
		if(!PRELOADED) {
			double2 v[NB * NB / 2 / POTRF_THREADS];
#pragma unroll
			for(int t = 0; t < NB * NB / 2 / POTRF_THREADS; ++ t) {
				const int e = tid + t * POTRF_THREADS, r = (e & (NB / 2 - 1)) * 2, c = e >> 6;
				v[t] = ld_blk2<COH>(Ablk + r + (int64_t)c * ld);
			}
#pragma unroll
			for(int t = 0; t < NB * NB / 2 / POTRF_THREADS; ++ t) {
				const int e = tid + t * POTRF_THREADS, r = (e & (NB / 2 - 1)) * 2, c = e >> 6;
				T[r + c * TS] = v[t].x;
				T[r + 1 + c * TS] = v[t].y;
			}
		}
		__syncthreads();
		SPP_STAMP(1, 0);
		if(tid < NB) {
			yv[tid] = (rhs_col >= 0 && tid < n_valid) ? T[tid + rhs_col * TS] : 0.0;
			if(!first_done || tid >= 16)
				dinv[tid] = 1.0;
		}
		__syncthreads();
		if(rhs_col >= 0 && tid < NB)
			T[tid + rhs_col * TS] = (tid == rhs_col) ? 1.0 : 0.0; // the rhs column becomes plain padding
		__syncthreads();
		if(wave == 0 && !first_done)
			diag_tile_factor<TS>(T, DvB, gd_slot(0), dinv, 0, lane, fail, info, k0);
		__syncthreads();
	}
	SPP_STAMP(2, 0);

	for(int J = 0; J < NB / 16; ++ J) {
		if(*fail) {
			if(pub.abort && tid == 0)
				__hip_atomic_store(pub.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		SPP_STAMP(3 + 6 * J, 0);
		const int j0 = J * 16;
		if(j0 >= n_valid) {
			// a panel of padding only (exact identity rows / columns, nothing right of them in these rows): its rows of R and
			// its columns of the inverse are unit vectors; nothing to eliminate, no LDS traffic, no barrier (uniform branch)
			const int nv16 = (n_valid + 15) & ~15;
			for(int e = tid; e < 16 * (NB - j0); e += POTRF_THREADS) {
				const int r = j0 + (e & 15), c = j0 + (e >> 4);
				if(r <= c && c != rhs_col)
					st_blk<WT>(&Ablk[r + (int64_t)c * ld], (r == c) ? 1.0 : 0.0);
			}
			if(HALF == 2) { // identity panel: its diagonal tile of the inverse is the identity; published like any other
				if(tid < 256) {
					const double v = ((tid & 15) == (tid >> 4)) ? 1.0 : 0.0;
					gd_slot(J)[(tid & 15) + (tid >> 4) * PT] = v;
					if(pub.dbuf)
						st_blk<1>(&pub.dbuf[256 * J + tid], v);
				}
				if(pub.flag) {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					lds_barrier();
					if(tid == 0) {
						fail[2] = 3 * (J + 1); // (the count of the streaming waves, kept in step)
						__hip_atomic_store(pub.flag, pub.base + J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				continue;
			}
			for(int e = tid; e < 16 * NB; e += POTRF_THREADS) {
				const int r = e & (NB - 1), c = j0 + (e >> 7);
				double v = (r == c) ? 1.0 : 0.0;
				if(r < c && HALF && (r >> 6) != (c >> 6))
					v = T[r + c * TS]; // R01 (zeros: a padding column)
				else if(r < c && r < nv16)
					v = T[c + r * TS]; // G[c][r] of the valid rows: zeros, assigned there
				if(WT < 2 || r <= c)
					st_blk<WT>(&tinv[r + c * NB], v);
			}
			continue;
		}
		const double *Dv = DvB + (J & 1) * 16 * PT, *Gd = gd_slot(J);
		// ---- B: row panel. tiles t < J: G part (columns 16 t ..), tiles t >= J: R part (columns 16 (t + 1) ..)
		if(wave < 7 && !(HALF == 1 && wave < (J & ~3)) && !(HALF == 2 && wave < J)) { // (HALF: no G tiles left of the panel's 64 x 64 block; 2: none)
			const int ct = (wave < J) ? wave : wave + 1;
			double *Y = T + j0 + (ct * 16) * TS;
			const v4f64 x = tile_atb(Dv, 1, PT, Y, 1, TS, lane); // X[i][j] = sum_k Dinv[k][i] Y[k][j]
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				Y[(l4 + 4 * r) + l15 * TS] = x[r];
		} else if(wave == 7 && lane < 16) { // the carried right-hand side, y_J = Dinv^T y_J
			double s = 0;
			for(int k = 0; k < 16; ++ k)
				s += Dv[k + lane * PT] * yv[j0 + k];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			yv[j0 + lane] = s;
		}
		SPP_STAMP(4 + 6 * J, 0);
		lds_barrier(); // LDS traffic only: the write-back stores stay in flight
		SPP_STAMP(5 + 6 * J, 0);
		SPP_STAMP(6 + 6 * J, 64);
		// ---- C (+ A of the next panel on wave 0): trailing update with the panel rows P = T[j0 .. j0 + 16, :]
		{
			const int nI = NB / 16 - 1 - J;         // row tiles I = J + 1 .. 7
			const int nR = nI * (nI + 1) / 2;       // R part: I <= K
			const int gI = HALF ? 3 - (J & 3) : nI;   // G part: rows I = J + 1 .. (HALF: the end of the panel's 64 x 64 block)
			const int gC = HALF ? (J & 3) + 1 : J + 1, gC0 = HALF ? (J & ~3) : 0; // columns Cb = gC0 .. J
			const int nG = (HALF == 2) ? 0 : gI * gC;
			// tile q -> (I, Ct, gpart); q = 0 is the next diagonal tile (I = K = J + 1)
			auto decode = [&](int q, int &I, int &Ct, bool &gpart) {
				gpart = q >= nR;
				if(!gpart) {
					int a = 0, rem = q;
					while(rem >= nI - a) {
						rem -= nI - a;
						++ a;
					}
					I = J + 1 + a;
					Ct = I + rem;
				} else {
					const int g = q - nR;
					I = J + 1 + g / gC;
					Ct = gC0 + g % gC;
				}
			};
			if(wave == 0) {
				if(nI > 0 && j0 + 16 < n_valid) { // (a next tile of padding only is not factored: its panel is skipped)
					const double *Pa = T + j0 + ((J + 1) * 16) * TS;
					const v4f64 d = tile_atb(Pa, 1, TS, Pa, 1, TS, lane);
					double *D = T + ((J + 1) * 16) + ((J + 1) * 16) * TS;
#pragma unroll
					for(int r = 0; r < 4; ++ r)
						D[(l4 + 4 * r) + l15 * TS] -= d[r];
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
					__builtin_amdgcn_wave_barrier();
					diag_tile_factor<TS>(T, DvB + ((J + 1) & 1) * 16 * PT, gd_slot(J + 1), dinv,
						j0 + 16, lane, fail, info, k0);
				}
			} else {
				// The waves that share wave 0's SIMD (4, 8, 12) stay out of the trailing update: the dependent MFMA
				// chain of the diagonal tile then has its matrix core to itself (7 000 -> 5 300 cycles per panel).
				// Row block J -- final since step B -- goes back to global memory inside this phase (below): R (rows
				// j0 .. j0+15 right of and on the diagonal) and columns j0 .. j0+15 of the inverse.
#ifndef SPP_POTRF_SIMD0_IDLE
#define SPP_POTRF_SIMD0_IDLE 1
#endif
#if SPP_POTRF_SIMD0_IDLE
				constexpr int NCW = NW - 1 - (NW / 4 - 1);
				const int slot = wave - 1 - (wave >> 2);
				const bool takes_tiles = (wave & 3) != 0;
#else
				static_assert(HALF != 2, "the streamed form puts its row tiles out through the waves that take no update tiles");
				constexpr int NCW = NW - 1;
				const int slot = wave - 1;
				const bool takes_tiles = true;
#endif
				if(HALF == 2 && pub.flag && !takes_tiles) {
					// streamed: row tile J of R is final since step B, and the three waves that take no update tiles put it out
					// NOW -- with the inverse of its diagonal tile -- instead of all waves after their tiles: each drains its own
					// stores, the last of the three to have done so stores the counter (an LDS count, no workgroup barrier):
					// the row tile is visible ~1.5 us after step B instead of after the whole of step C
					const int t3 = ((wave >> 2) - 1) * 64 + lane; // 0 .. 191
					for(int e = t3; e < 16 * (NB - j0); e += 3 * 64) {
						const int r = j0 + (e & 15), c = j0 + (e >> 4);
						if(r <= c && c != rhs_col)
							st_blk<1>(&Ablk[r + (int64_t)c * ld], T[r + c * TS]);
					}
					if(pub.dbuf)
						for(int e = t3; e < 256; e += 3 * 64) // Dinv[k][i] at k + 16 i
							st_blk<1>(&pub.dbuf[256 * J + e], Dv[(e & 15) + (e >> 4) * PT]);
					// (no wait here: a write-through store takes 2-3 us to drain, longer than wave 0's elimination of the next
					// tile -- waiting inside the panel held its closing barrier up by up to 1 500 cycles in the first panels.
					// The three drain and store the counter right behind that barrier: streamed_publish below)
				}
				for(int q = (takes_tiles ? slot + 1 : (1 << 20)); q < nR + nG; q += 2 * NCW) { // two tiles per wave and round
					const int q1 = q + NCW;
					int I0, C0, I1 = 0, C1 = 0;
					bool g0, g1 = false;
					decode(q, I0, C0, g0);
					const bool have1 = q1 < nR + nG;
					if(have1)
						decode(q1, I1, C1, g1);
					const double *a0 = T + j0 + (I0 * 16) * TS;
					const bool gd0 = g0 && C0 == J;
					const double *b0 = gd0 ? Gd : T + j0 + (C0 * 16) * TS;
					if(have1) {
						const double *a1 = T + j0 + (I1 * 16) * TS;
						const bool gd1 = g1 && C1 == J;
						const double *b1 = gd1 ? Gd : T + j0 + (C1 * 16) * TS;
						v4f64 d0, d1;
						tile_atb2<TS>(a0, b0, 1, gd0 ? PT : TS, a1, b1, 1, gd1 ? PT : TS, lane, d0, d1);
						double *D0 = T + (I0 * 16) + (C0 * 16) * TS, *D1 = T + (I1 * 16) + (C1 * 16) * TS;
#pragma unroll
						for(int r = 0; r < 4; ++ r) {
							double *dp = D0 + (l4 + 4 * r) + l15 * TS;
							*dp = gd0 ? -d0[r] : *dp - d0[r]; // first touch of a G tile: assign
							double *dq = D1 + (l4 + 4 * r) + l15 * TS;
							*dq = gd1 ? -d1[r] : *dq - d1[r];
						}
					} else {
						const v4f64 d0 = tile_atb(a0, 1, TS, b0, 1, gd0 ? PT : TS, lane);
						double *D0 = T + (I0 * 16) + (C0 * 16) * TS;
#pragma unroll
						for(int r = 0; r < 4; ++ r) {
							double *dp = D0 + (l4 + 4 * r) + l15 * TS;
							*dp = gd0 ? -d0[r] : *dp - d0[r];
						}
					}
				}
				// write-back of row block J by all of waves 1..15, after their tiles (the stores stay in flight across
				// the barriers below). Measured alternatives: waves 4 / 8 / 12 alone doing it (they have no tiles) took
				// longer than wave 0's elimination and slowed it through the shared SIMD; three tiles per wave and round
				// cost what two rounds of two cost (the update is bound by the MFMA pipe: one v_mfma_f64_16x16x4 per
				// ~105 cycles and SIMD, which is also what bounds the bare-MFMA loop at 47 TFLOP/s).
				if(!(HALF == 2 && pub.flag)) {
					const int t15 = (wave - 1) * 64 + lane; // 0 .. 959
					for(int e = t15; e < 16 * (NB - j0); e += (NW - 1) * 64) { // R: 16 rows x (NB - j0) columns
						const int r = j0 + (e & 15), c = j0 + (e >> 4);
						if(r <= c && c != rhs_col)
							st_blk<WT>(&Ablk[r + (int64_t)c * ld], T[r + c * TS]);
					}
					for(int e = t15; HALF != 2 && e < 16 * NB; e += (NW - 1) * 64) { // inverse: columns j0 .. j0+15, all 128 rows
						const int r = e & (NB - 1), c = j0 + (e >> 7);
						double v = 0;
						if(r == c)
							v = dinv[r];
						else if(r < c)
							v = (HALF && (r >> 6) != (c >> 6)) ? T[r + c * TS] : T[c + r * TS]; // R01[r][c] (final since panel 3) : G[c][r]
						if(WT < 2 || r <= c)
							st_blk<WT>(&tinv[r + c * NB], v);
					}
				}
				// rhs: y_i -= sum_k P[k][i] y_J[k], by the threads of waves 8..9
				const int ti = tid - 512;
				if(ti >= j0 + 16 && ti < NB) {
					double s = 0;
					for(int k = 0; k < 16; ++ k)
						s += T[(j0 + k) + ti * TS] * yv[j0 + k];
					yv[ti] -= s;
				}
			}
		}
		SPP_STAMP(7 + 6 * J, 0);
		SPP_STAMP(8 + 6 * J, 64);
		lds_barrier(); // LDS traffic only: the write-back stores stay in flight
#if SPP_POTRF_SIMD0_IDLE
		if(HALF == 2 && pub.flag && wave > 0 && (wave & 3) == 0) {
			// streamed_publish: each of the three waves that put row tile J out drains its own stores; the last of them to
			// have done so stores the counter (an LDS count, no workgroup barrier)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			int old_cnt = 0;
			if(lane == 0)
				old_cnt = __hip_atomic_fetch_add(&fail[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if(lane == 0 && old_cnt == 3 * J + 2)
				__hip_atomic_store(pub.flag, pub.base + J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
#endif
	}
	SPP_STAMP(51, 0);
	if(*fail) {
		if(pub.abort && tid == 0)
			__hip_atomic_store(pub.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	if(HALF == 2) {
		// ---- the inverse, after the factorization: G = R^-T (lower triangular) in the unused lower triangle of the image,
		// by recursive doubling over the 16 x 16 tiles. With the diagonal tiles G_JJ known (the tile factorization left
		// them in the Gd slots), a block [[G11, 0], [G21, G22]] of twice the size has G21 = -G22 (L21 G11), L21 = R12^T:
		// two stages of independent tile products per level (M = L21 G11 into the destination tiles, then G21 from M),
		// 4, 8, 16 tiles at the three levels: 21 k cycles, behind the last published row tile.
		lds_barrier();
#pragma unroll 1
		for(int h = 1; h < NB / 16; h *= 2) { // half size of the level's blocks, in tiles
			const int nt = 4 * h;               // tiles of this level: (8 / 2h) blocks x h x h
			const int blk = wave / (h * h), within = wave % (h * h);
			const int base = blk * 2 * h, I = base + h + within / h, K = base + within % h;
			const bool mine = wave < nt;
			double *D = T + 16 * I + (16 * K) * TS; // destination tile (I, K), I > K
			if(mine) { // M(I, K) = sum_{Jt = K .. base+h-1} R_JI^T G_JK
				v4f64 acc = (v4f64){0, 0, 0, 0};
				for(int Jt = K; Jt < base + h; ++ Jt) {
					const double *a = T + 16 * Jt + (16 * I) * TS; // R_JI: element (k, i) at a[k + i TS]
					const v4f64 m = (Jt == K) ? tile_atb(a, 1, TS, gd_slot(K), 1, PT, lane)
					                          : tile_atb(a, 1, TS, T + 16 * Jt + (16 * K) * TS, 1, TS, lane);
					acc += m;
				}
#pragma unroll
				for(int r = 0; r < 4; ++ r)
					D[(l4 + 4 * r) + l15 * TS] = acc[r];
			}
			lds_barrier();
			v4f64 res = (v4f64){0, 0, 0, 0};
			if(mine) { // G(I, K) = -sum_{Jt = base+h .. I} G_IJ M(J, K); A operand element (k, i) = G_IJ[i][k]
				for(int Jt = base + h; Jt <= I; ++ Jt) {
					const double *b = T + 16 * Jt + (16 * K) * TS; // M(J, K)
					const v4f64 g = (Jt == I) ? tile_atb(gd_slot(I), PT, 1, b, 1, TS, lane)
					                          : tile_atb(T + 16 * I + (16 * Jt) * TS, TS, 1, b, 1, TS, lane);
					res += g;
				}
			}
			lds_barrier(); // every M tile of the level has been read
			if(mine) {
#pragma unroll
				for(int r = 0; r < 4; ++ r)
					D[(l4 + 4 * r) + l15 * TS] = -res[r];
			}
			lds_barrier();
		}
		// the whole inverse goes back: tinv(r, c) = G[c][r] for r < c, 1 / R_cc on the diagonal, zeros below
		for(int e = tid; e < NB * NB; e += POTRF_THREADS) {
			const int r = e & (NB - 1), c = e >> 7;
			if(WT < 2 || r <= c)
				st_blk<0>(&tinv[r + c * NB], (r == c) ? dinv[r] : ((r < c) ? T[c + r * TS] : 0.0));
		}
	}
	// R and the inverse went back block row by block row inside the loop (HALF = 2: the inverse just now); the carried rhs remains
	if(rhs_col >= 0 && tid < n_valid)
		st_blk<WT>(&Ablk[tid + (int64_t)rhs_col * ld], yv[tid]);
	SPP_STAMP(52, 0);
}


} // namespace spp
