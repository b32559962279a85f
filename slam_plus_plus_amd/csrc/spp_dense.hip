// spp_dense.hip -- dense upper Cholesky (R^T R = S) + solves for the reduced camera system and for
// large frontal matrices, gfx950 only.
//
// Replaces (functionally) the reference's dense reduced solve
//   CLinearSolver_DenseEigen::Solve_PosDef   src/slam/LinearSolver_Schur.cpp:2314-2331
//   (Convert_to_Dense + Eigen::LLT<MatrixXd, Upper> + solve), and the CULA call it stands in for
//   src/slam/LinearSolver_Schur_GPU.cpp:759 (culaDevicePosv) -- as a spec only, nothing is ported.
//
// Design (DESIGN.md section "dense factor"):
//   * S is n_pad x n_pad column-major (ld = n_pad, a multiple of NB = 128, and n_pad > n so that at
//     least one padding column exists). Only the upper triangle is read (the reference zero-fills
//     the lower one, BlockMatrix.cpp:9207-9239). Padding diagonal = 1.
//   * The right-hand side lives in padding column n of S: the TRSM / trailing-update kernels then
//     perform the forward substitution R^T y = b for free (b_i -= R_ki^T y_k is the same GEMM).
//   * Right-looking, block size 128:
//       potrf_diag   one workgroup, block in LDS, square-root-free elimination with ONE barrier per
//                    pivot; the unused lower triangle of the block accumulates (R_kk^-1)^T so the
//                    panel solve becomes a GEMM,
//       trsm         R_kj = (R_kk^-1)^T S_kj   as an in-place MFMA GEMM (128 x 32 tiles),
//       syrk         S_ij -= R_ki^T R_kj       MFMA f64 16x16x4 GEMM, 128 x 128 tiles, upper tiles only.
//     The trailing update is the dominant kernel: n^3/3 flops, bounded by the fp64 MFMA roofline.
//   * Backward substitution R x = y: one launch per block column, coalesced column-oriented GEMV.
//
// MFMA f64 fragment maps used (cdna_hip_programming.md section 3, v_mfma_f64_16x16x4_f64):
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15]
//   D: 4 doubles per lane, D[row = (l >> 4) + 4 r][col = l & 15], r = 0..3.
// The operands are fed swapped (A_op <- B tile, B_op <- A tile) so that the 16 lanes l & 15 hold 16
// CONSECUTIVE ROWS of C: every store/load of C touches whole 128-byte segments of a column.

#include "spp_internal.h"
#include "spp_tiles.h"
#include "spp_dense_dev.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <mutex>

namespace spp {


constexpr int BK = 16;           // k-slab staged through LDS
constexpr int LDS_STRIDE = 18;   // doubles per tile column in LDS: conflict-free ds_read_b64, 16-B aligned

// --------------------------------------------------------------------------------------------------
// C (M x N) {-=, =} A^T B,  A: K x M (lda), B: K x N (ldb), K % 16 == 0. Column-major.
// MODE 0: C -= A^T B (trailing update);  MODE 1: C = A^T B (may alias B when BM covers all rows).
// --------------------------------------------------------------------------------------------------
// MINW = waves per SIMD the register allocator must leave room for (2nd argument of
// __launch_bounds__): 2 for the 512-thread configuration = one workgroup per CU and a 256-VGPR budget
// (at the default budget of 128 the staging registers were spilled to scratch inside the k-loop)
// one BM x BN tile of C at (m0, n0) by the whole workgroup (the body shared by the kernels below)
// SC1: the C tile is stored write-through (agent-scope atomic = sc1 stores): a tile another workgroup reads behind a
// counter hand-off, without a release fence (lookahead schedule, the first tile rows of a bulk update)
template <int BM, int BN, int WM, int WN, int MODE, int DEPTH, int BKT, int SC1 = 0>
__device__ __forceinline__ void gemm_tn_tile(const int64_t m0, const int64_t n0, int64_t M, int64_t N, int K,
	const double *__restrict__ A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, double *gemm_lds)
{
	constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
	constexpr int TA = WM / 16, TB = WN / 16;       // MFMA tiles per wave
	constexpr int PPC = BKT / 2, LSTR = BKT + 2;            // 16-byte pieces per column, LDS column stride
	constexpr int PA = (BM * PPC) / NT, PB = (BN * PPC) / NT; // pieces per thread per slab
	static_assert((BM * PPC) % NT == 0 && (BN * PPC) % NT == 0, "tile/threads mismatch");

	double *As = gemm_lds, *Bs = gemm_lds + BM * LSTR;

	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave % NWM) * WM, wn = (wave / NWM) * WN; // wave-uniform (scalar registers)
	const int l15 = lane & 15, l4 = lane >> 4;
	// C(m, n) = (uniform tile corner) + lane part: one 32-bit offset register serves all 16 accesses
	const uint32_t c_lane = (uint32_t)(l15 + l4 * ldc);

	v4f64 acc[TB][TA];
	// MODE 0: the old C values are fetched up front (their latency hides under the whole k-loop) and
	// folded in as the initial accumulator: acc = -C, C_new = -(acc + A^T B)
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a) {
			acc[b][a] = (v4f64){0, 0, 0, 0};
			if(MODE == 0) {
				const int64_t mu = m0 + wm + a * 16, nu = n0 + wn + b * 16;
				const double *Cu = C + mu + nu * ldc;
#pragma unroll
				for(int r = 0; r < 4; ++ r) {
					if(mu + l15 < M && nu + l4 + 4 * r < N)
						acc[b][a][r] = -(Cu + (int64_t)(4 * r) * ldc)[c_lane];
				}
			}
		}

	// global staging: piece p -> (column p / 8, 16-byte piece p % 8); two slabs in flight
	double2 ra0[PA], ra1[PA], rb0[PB], rb1[PB];
	uint32_t pa[PA], pb[PB]; // element offsets (32-bit: fewer VGPRs than pointers)
#pragma unroll
	for(int i = 0; i < PA; ++ i) {
		int p = tid + i * NT;
		int64_t col = m0 + (p / PPC);
		if(col > M - 1) col = M - 1; // clamp: computes garbage that is never stored
		pa[i] = (uint32_t)(col * lda + (p % PPC) * 2);
	}
#pragma unroll
	for(int i = 0; i < PB; ++ i) {
		int p = tid + i * NT;
		int64_t col = n0 + (p / PPC);
		if(col > N - 1) col = N - 1;
		pb[i] = (uint32_t)(col * ldb + (p % PPC) * 2);
	}
#define SPP_LOAD_SLAB(buf, kofs) \
	_Pragma("unroll") for(int i = 0; i < PA; ++ i) (buf ? ra1 : ra0)[i] = *(const double2*)(A + (size_t)pa[i] + (kofs)); \
	_Pragma("unroll") for(int i = 0; i < PB; ++ i) (buf ? rb1 : rb0)[i] = *(const double2*)(B + (size_t)pb[i] + (kofs));
#define SPP_STORE_SLAB(buf) \
	_Pragma("unroll") for(int i = 0; i < PA; ++ i) { int p = tid + i * NT; \
		*(double2*)(&As[(p / PPC) * LSTR + (p % PPC) * 2]) = (buf ? ra1 : ra0)[i]; } \
	_Pragma("unroll") for(int i = 0; i < PB; ++ i) { int p = tid + i * NT; \
		*(double2*)(&Bs[(p / PPC) * LSTR + (p % PPC) * 2]) = (buf ? rb1 : rb0)[i]; }
#define SPP_COMPUTE_SLAB() \
	_Pragma("unroll") for(int kk = 0; kk < BKT / 4; ++ kk) { \
		double fa[TA], fb[TB]; \
		_Pragma("unroll") for(int a = 0; a < TA; ++ a) fa[a] = As[(wm + a * 16 + l15) * LSTR + kk * 4 + l4]; \
		_Pragma("unroll") for(int b = 0; b < TB; ++ b) fb[b] = Bs[(wn + b * 16 + l15) * LSTR + kk * 4 + l4]; \
		_Pragma("unroll") for(int b = 0; b < TB; ++ b) \
			_Pragma("unroll") for(int a = 0; a < TA; ++ a) \
				acc[b][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[b][a], 0, 0, 0); \
	}

	if(DEPTH == 1) {
		// one slab in flight (fewer staging registers: no scratch at 128 VGPRs, two workgroups per CU)
		SPP_LOAD_SLAB(0, 0)
		for(int k0 = 0; k0 < K; k0 += BKT) {
			__syncthreads();
			SPP_STORE_SLAB(0)
			__syncthreads();
			if(k0 + BKT < K) {
				SPP_LOAD_SLAB(0, k0 + BKT)
			}
			SPP_COMPUTE_SLAB()
		}
	} else {
	SPP_LOAD_SLAB(0, 0)
	if(BKT < K) {
		SPP_LOAD_SLAB(1, BKT)
	}
	for(int k0 = 0; k0 < K; k0 += 2 * BKT) {
		__syncthreads(); // previous slab fully consumed
		SPP_STORE_SLAB(0)
		__syncthreads();
		if(k0 + 2 * BKT < K) {
			SPP_LOAD_SLAB(0, k0 + 2 * BKT)
		}
		SPP_COMPUTE_SLAB()
		if(k0 + BKT < K) {
			__syncthreads();
			SPP_STORE_SLAB(1)
			__syncthreads();
			if(k0 + 3 * BKT < K) {
				SPP_LOAD_SLAB(1, k0 + 3 * BKT)
			}
			SPP_COMPUTE_SLAB()
		}
	}
	}
#undef SPP_LOAD_SLAB
#undef SPP_STORE_SLAB
#undef SPP_COMPUTE_SLAB
	if(MODE == 1)
		__syncthreads(); // in-place: every wave has finished reading B before anyone writes C

	// D[row = l4 + 4 r][col = l15] of tile (b, a) is C(m = wm + 16 a + l15, n = wn + 16 b + l4 + 4 r)
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a) {
			const int64_t mu = m0 + wm + a * 16, nu = n0 + wn + b * 16;
			double *Cu = C + mu + nu * ldc;
#pragma unroll
			for(int r = 0; r < 4; ++ r) {
				if(mu + l15 < M && nu + l4 + 4 * r < N)
				{
					const double val = (MODE == 0) ? -acc[b][a][r] : acc[b][a][r];
					if(SC1)
						__hip_atomic_store(&(Cu + (int64_t)(4 * r) * ldc)[c_lane], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					else
						(Cu + (int64_t)(4 * r) * ldc)[c_lane] = val;
				}
			}
		}
}

template <int BM, int BN, int WM, int WN, int MODE, int DEPTH = 2, int MINW = 1, int BKT = 16>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) __attribute__((amdgpu_waves_per_eu(MINW == 2 ? 2 : 1, MINW == 2 ? 2 : 8)))
void gemm_tn_kernel(int64_t M, int64_t N, int K, const double *__restrict__ A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, int upper_only)
{
	const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
	if(upper_only && m0 >= n0 + BN)
		return; // tile strictly below the diagonal
	extern __shared__ double gemm_lds[];
	gemm_tn_tile<BM, BN, WM, WN, MODE, DEPTH, BKT>(m0, n0, M, N, K, A, lda, B, ldb, C, ldc, gemm_lds);
}

// --------------------------------------------------------------------------------------------------
// Trailing update with a fine-grained tail. A 128 x 128 tile occupies its workgroup for ~55 us and the
// chip holds 512 of them: 528 tiles took as long as 1024. Here the upper tiles are enumerated in ONE
// dimension (tile columns, rows ascending); the first n128 are processed whole, every remaining tile is
// cut into four 64 x 64 quarters, each by its own workgroup (16 waves, one MFMA tile per wave): the
// dispatcher hands out the long items first and levels the end of the launch with the short ones.
//   nt = tile rows (M), ntc = tile columns (N >= M); tile t -> (i, j), i <= min(j, nt - 1)
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void upper_tile_of(int64_t t, int nt, int &ti, int &tj)
{
	const int64_t tri = (int64_t)nt * (nt + 1) / 2;
	if(t >= tri) { // rectangular part right of the square (right-hand side columns)
		const int64_t q = t - tri;
		tj = nt + (int)(q / nt);
		ti = (int)(q % nt);
		return;
	}
	int j = (int)((__dsqrt_rn(8.0 * (double)t + 1.0) - 1.0) * 0.5);
	while((int64_t)j * (j + 1) / 2 > t) -- j;
	while((int64_t)(j + 1) * (j + 2) / 2 <= t) ++ j;
	tj = j;
	ti = (int)(t - (int64_t)j * (j + 1) / 2);
}

} // namespace spp
#include "spp_dense_444.h" // the 128 x 128 tile built around v_mfma_f64_4x4x4_4b (LDS-DMA staging, swizzled images)
namespace spp {

#ifndef SPP_MIXED_WAVES
#define SPP_MIXED_WAVES 8 // waves per SIMD the register allocator leaves room for: 8 = two workgroups per CU at 64 VGPRs, 4 = one at 128
#endif
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(SPP_MIXED_WAVES, SPP_MIXED_WAVES)))
void gemm_tn_mixed_kernel(int64_t M, int64_t N, int K, const double *__restrict__ A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, int nt, int64_t n128, int use444)
{
	extern __shared__ __attribute__((aligned(16))) double gemm_lds[];
	const int64_t b = blockIdx.x;
	int ti, tj;
	if(b < n128) {
		upper_tile_of(b, nt, ti, tj);
		if(use444 && (int64_t)(ti + 1) * 128 <= M && (int64_t)(tj + 1) * 128 <= N) // (edge tiles: the bounds-checked tile)
		{
			if(use444 == 2)
				gemm_tn_tile_444<0>((int64_t)ti * 128, (int64_t)tj * 128, K, A, lda, B, ldb, C, ldc, gemm_lds);
			else
				gemm_tn_tile_dma<0>((int64_t)ti * 128, (int64_t)tj * 128, K, A, lda, B, ldb, C, ldc, gemm_lds);
		} else
			gemm_tn_tile<128, 128, 32, 32, 0, 1, 16>((int64_t)ti * 128, (int64_t)tj * 128, M, N, K, A, lda, B, ldb, C, ldc, gemm_lds);
	} else {
		const int64_t q = b - n128;
		upper_tile_of(n128 + (q >> 2), nt, ti, tj);
		const int64_t m0 = (int64_t)ti * 128 + (q & 1) * 64, n0 = (int64_t)tj * 128 + ((q >> 1) & 1) * 64;
		if(m0 >= n0 + 64 || m0 >= M || n0 >= N)
			return; // quarter strictly below the diagonal or outside the matrix
		gemm_tn_tile<64, 64, 16, 16, 0, 1, 32>(m0, n0, M, N, K, A, lda, B, ldb, C, ldc, gemm_lds);
	}
}

// returns false when the shape is not worth it (the caller falls back to the plain kernels)
static bool launch_gemm_mixed(hipStream_t s, int64_t M, int64_t N, int K, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc)
{
	const int64_t slots = 512; // concurrently resident 128 x 128 workgroups (2 per CU)
	if(K % 32 != 0 || N < M)
		return false;
	const int64_t nt = (M + 127) / 128, ntc = (N + 127) / 128;
	const int64_t T = nt * (nt + 1) / 2 + (ntc - nt) * nt;
	const int64_t n128 = (T / slots) * slots;
	const int64_t grid = n128 + 4 * (T - n128);
	static int use444 = -1;
	if(use444 < 0) {
		const char *e = getenv("SPP_TILE_444"); // whole 128 x 128 tiles: 0 register-staged 16x16x4 tile (rounds 1-2, default), 1 LDS-DMA 16x16x4 tile, 2 LDS-DMA 4x4x4_4b tile
		use444 = e ? atoi(e) : 0;
	}
	static uint64_t mixed_attr_seen = 0;
	if(first_on_this_device(mixed_attr_seen))
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			T444_LDS_DOUBLES * (int)sizeof(double)));
	const size_t lds = use444 ? (size_t)T444_LDS_DOUBLES * sizeof(double)
		: (size_t)(128 + 128) * 18 * sizeof(double); // >= (64 + 64) * 34 doubles of the quarter path
	hipLaunchKernelGGL(gemm_tn_mixed_kernel, dim3((unsigned)grid), dim3(1024), lds, s,
		M, N, K, A, lda, B, ldb, C, ldc, (int)nt, n128, use444);
	return true;
}

template <int BM, int BN, int WM, int WN, int MODE, int DEPTH = 2, int MINW = 1, int BKT = 16>
static void launch_gemm(hipStream_t s, int64_t M, int64_t N, int K, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only)
{
	dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
	dim3 block((BM / WM) * (BN / WN) * 64);
	if(!grid.x || !grid.y)
		return;
	const size_t lds = (size_t)(BM + BN) * (BKT + 2) * sizeof(double);
	static uint64_t attr_seen = 0;
	if(lds > 65536 && first_on_this_device(attr_seen)) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_kernel<BM, BN, WM, WN, MODE, DEPTH, MINW, BKT>,
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	}
	SPP_REQUIRE(K % BKT == 0, SPP_E_BADARG, "gemm: K must be a multiple of the slab depth");
	hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN, MODE, DEPTH, MINW, BKT>), grid, block, lds, s,
		M, N, K, A, lda, B, ldb, C, ldc, upper_only ? 1 : 0);
}

// --------------------------------------------------------------------------------------------------
// Fully staged variant for the latency-critical chain kernels (row-panel solve and tile-row update):
// K <= 128, both operand panels are loaded into LDS ONCE (all global loads in flight together, one
// barrier), then the MFMAs run without further synchronization. LDS row stride K_MAX + 2 doubles:
// conflict-free ds_read_b64 fragments (lanes l & 15 step 4 banks, lanes l >> 4 step 2 banks).
// --------------------------------------------------------------------------------------------------
// (FS_KMAX, FS_STRIDE and the staged tile product itself: spp_dense_dev.h)

// ---- cross-stream hand-offs through words in device memory ------------------------------------------
// A flag holds the epoch of the factorization that last completed a piece of work. Signalling is a
// one-wave kernel BEHIND the producing kernel on the producer's stream, waiting a one-wave kernel IN FRONT
// of the consumer on its stream: the kernel boundaries are the release and the acquire (no fences inside the
// compute kernels), and a waiter holds one wave slot, never the slots the producer still needs (a consumer
// that polled in its own prologue would park hundreds of workgroups on the CUs of the kernel it waits for).
// Every wait is bounded: on a timeout (100 MHz wall clock) the abort word is set, all later waits fall
// through and the host reports the failure.
struct FlagWait {
	const int *flag; // nullptr: no wait
	int value;
	int *abort;
	long long timeout_ticks;
};

__device__ __forceinline__ void flag_spin(const FlagWait &fw)
{
	const long long t0 = wall_clock64();
	for(int it = 0;; ++ it) {
		if(__hip_atomic_load(fw.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fw.value)
			return;
		if((it & 15) == 15) {
			if(__hip_atomic_load(fw.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
				return;
			if(wall_clock64() - t0 > fw.timeout_ticks) {
				__hip_atomic_store(fw.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
		}
		__builtin_amdgcn_s_sleep(2);
	}
}

__global__ void flag_signal_kernel(int *flag, int value)
{
	if(threadIdx.x == 0)
		__hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flag_wait_kernel(FlagWait fw)
{
	if(threadIdx.x == 0)
		flag_spin(fw);
}

// signal one flag, then wait for another: the chain stream's "row panel k is complete" and "bulk update k-1 is
// complete" sit next to each other in its order -- one launch instead of two
__global__ void flag_signal_wait_kernel(int *flag, int value, FlagWait fw)
{
	if(threadIdx.x == 0) {
		__hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		flag_spin(fw);
	}
}

template <int BM, int BN, int WM, int WN, int MODE, int ATRI = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64)
void gemm_tn_staged_kernel(int64_t M, int64_t N, int K, const double *__restrict__ A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, int upper_only)
{
	extern __shared__ double fs_lds[];
	const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
	if(upper_only && m0 >= n0 + BN)
		return;
	gemm_tn_staged_tile<BM, BN, WM, WN, MODE, ATRI>(m0, n0, M, N, A, lda, B, ldb, C, ldc, fs_lds);
}

template <int BM, int BN, int WM, int WN, int MODE, int ATRI = 0>
static void launch_gemm_staged(hipStream_t s, int64_t M, int64_t N, int K, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only)
{
	static uint64_t attr_seen = 0;
	const size_t lds = (size_t)(BM + BN) * FS_STRIDE * sizeof(double);
	if(first_on_this_device(attr_seen)) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tn_staged_kernel<BM, BN, WM, WN, MODE, ATRI>,
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	}
	SPP_REQUIRE(K == FS_KMAX, SPP_E_BADARG, "staged gemm: K must be 128");
	dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
	dim3 block((BM / WM) * (BN / WN) * 64);
	if(!grid.x || !grid.y)
		return;
	SPP_REQUIRE(!ATRI || M <= BM, SPP_E_BADARG, "staged gemm: a triangular A needs one tile row");
	hipLaunchKernelGGL((gemm_tn_staged_kernel<BM, BN, WM, WN, MODE, ATRI>), grid, block, lds, s,
		M, N, K, A, lda, B, ldb, C, ldc, upper_only ? 1 : 0);
}

bool dense_gemm_tn_sub(spp_ctx *ctx, int64_t m, int64_t n, int64_t k, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only)
{
	SPP_REQUIRE(k % BK == 0, SPP_E_BADARG, "gemm_tn_sub: k must be a multiple of 16");
	if(!m || !n || !k)
		return false;
	// 128 x 128 tiles when they fill the chip, 64 x 64 tiles for the tail of the factorization; an upper update of at
	// least 50 tiles goes through the mixed-granularity kernel (whole tiles first, the tail of the launch in quarters).
	// (Measured and dropped in rounds 1-2: 64 x 32 / 32 x 64 wave tiles, two slabs in flight, 32- and 64-deep slabs,
	// 128 x 64 and 256 x 128 workgroup tiles, an LDS pad as occupancy limiter -- DESIGN.md section 3.)
	int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128);
	if(upper_only)
		t128 = t128 / 2 + 1;
	bool big = false;
	if(upper_only && t128 >= 50 && launch_gemm_mixed(ctx->stream, m, n, (int)k, A, lda, B, ldb, C, ldc))
		big = true;
	else if(t128 >= 192) {
		launch_gemm<128, 128, 32, 32, 0, 1, 1>(ctx->stream, m, n, (int)k, A, lda, B, ldb, C, ldc, upper_only);
		big = true;
	} else
		launch_gemm<64, 64, 32, 32, 0>(ctx->stream, m, n, (int)k, A, lda, B, ldb, C, ldc, upper_only);
	SPP_HIP_CHECK(hipGetLastError());
	return big; // true: a 16-wave 128 x 128-tile kernel (the one the roofline is reported for) was launched
}

__global__ __launch_bounds__(POTRF_THREADS)
void potrf_diag_kernel(double *__restrict__ Ablk, int64_t ld, int n_valid, int has_rhs,
	double *__restrict__ tinv, int *__restrict__ info, int64_t k0)
{
	extern __shared__ double sm[];
	potrf_diag_body<false, 0, 1>(Ablk, ld, n_valid, has_rhs, tinv, info, k0, sm);
}

// row panel of a step: R_kj = R_kk^-T S_kj in place, one 16-column slab per workgroup (panel_solve_slab: the diagonal
// block's inverse in the two-halves form potrf_diag_body<.., .., 1> stores)
constexpr int PANEL_THREADS = 512;
__global__ __launch_bounds__(PANEL_THREADS)
void panel_solve_kernel(int64_t N, const double *__restrict__ tinv, double *Y, int64_t ld)
{
	extern __shared__ double fs_lds[];
	panel_solve_slab<PANEL_THREADS>((int64_t)blockIdx.x * 16, N, tinv, Y, ld, fs_lds);
}

static void launch_panel_solve(hipStream_t s, int64_t N, const double *tinv, double *Y, int64_t ld)
{
	static uint64_t attr_seen = 0;
	const size_t lds = (size_t)(NB + 16) * FS_STRIDE * sizeof(double);
	if(first_on_this_device(attr_seen))
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)panel_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	if(N <= 0)
		return;
	hipLaunchKernelGGL(panel_solve_kernel, dim3((unsigned)((N + 15) / 16)), dim3(PANEL_THREADS), lds, s, N, tinv, Y, ld);
}

// The two-halves form [T0 R01; 0 T1] of every stored diagonal-block inverse becomes the full inverse
// [T0  -T0 R01 T1; 0  T1] in place -- once per factorization, off its critical chain, for the backward substitution
// (which multiplies by whole block inverses). One workgroup per block, one 16 x 16 tile of the 64 x 64 block per wave.
constexpr int TF_STRIDE = 66;
__global__ __launch_bounds__(1024)
void tinv_finish_kernel(double *__restrict__ tinv_all)
{
	__shared__ double T0[64 * TF_STRIDE], R01[64 * TF_STRIDE], T1[64 * TF_STRIDE], Nm[64 * TF_STRIDE];
	double *tv = tinv_all + (size_t)blockIdx.x * NB * NB;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
	for(int e = tid; e < 64 * 64; e += 1024) {
		const int r = e & 63, c = e >> 6;
		T0[r + c * TF_STRIDE] = tv[r + (size_t)c * NB];
		R01[r + c * TF_STRIDE] = tv[r + (size_t)(64 + c) * NB];
		T1[r + c * TF_STRIDE] = tv[64 + r + (size_t)(64 + c) * NB];
	}
	__syncthreads();
	const int it = wave & 3, jt = wave >> 2; // tile (it, jt) of the 64 x 64 block
	{
		// N = R01 T1 (T1 upper triangular: k tiles 0 .. jt)
		v4f64 acc = (v4f64){0, 0, 0, 0};
		for(int kt = 0; kt <= jt; ++ kt) {
			const v4f64 m = tile_atb(R01 + 16 * it + 16 * kt * TF_STRIDE, TF_STRIDE, 1, T1 + 16 * kt + 16 * jt * TF_STRIDE, 1, TF_STRIDE, lane);
			acc += m;
		}
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			Nm[16 * it + l4 + 4 * r + (16 * jt + l15) * TF_STRIDE] = acc[r];
	}
	__syncthreads();
	{
		// -T0 N (T0 upper triangular: k tiles it .. 3)
		v4f64 acc = (v4f64){0, 0, 0, 0};
		for(int kt = it; kt < 4; ++ kt) {
			const v4f64 m = tile_atb(T0 + 16 * it + 16 * kt * TF_STRIDE, TF_STRIDE, 1, Nm + 16 * kt + 16 * jt * TF_STRIDE, 1, TF_STRIDE, lane);
			acc += m;
		}
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			tv[16 * it + l4 + 4 * r + (size_t)(64 + 16 * jt + l15) * NB] = -acc[r];
	}
}


// --------------------------------------------------------------------------------------------------
// Fused chain kernel: trailing update from ONE row panel + factorization of the next diagonal block.
//
// The chain of the factorization used to run  tile row (20 us) -> potrf_diag (30 us) -> panel solve (6 us)  one after
// the other, although potrf_diag needs ONE 128 x 128 tile of the tile row. Here both are one launch:
//   region  C = rows [r0, r0 + M) x cols [r0, r0 + N) of the matrix (upper part), panel P = 128 rows at kp0
//   blockIdx.x in [0, nA)  32 x 32 sub-tiles of the region's FIRST 128 x 128 tile (upper ones), four waves each (the
//                          other twelve leave at once); each signals a counter behind an agent-scope release
//   blockIdx.x == nA       (do_potrf) waits for the nA signals -- producers have smaller block ids, are dispatched
//                          first and wait for nobody; the spin is bounded all the same --, acquires, factors the tile
//                          (potrf_diag_body) while
//   blockIdx.x  > nA       the rest of the region is updated in 64 x 64 blocks (16 waves, operands fully staged).
// LDS is sized for the factorization (143 KB): one workgroup per CU. The panel solve of the next step follows as its
// own launch (it needs the whole tile row AND the factor).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(POTRF_THREADS)
void update_potrf_kernel(int64_t M, int64_t N, const double *P, int64_t ld, double *C, int nA, int target, int do_potrf,
	int n_valid, int has_rhs, double *tinv, int *info, int64_t k0_next, int *counter, int *abort, long long timeout_ticks)
{
	extern __shared__ double sm[];
	const int b = (int)blockIdx.x, tid = threadIdx.x;
	if(b < nA) {
		if(tid >= 256)
			return; // four waves per sub-tile (a barrier does not wait for waves that have ended)
		// sub-tile b -> (i, j), i <= j, of the first tile, columns first
		const int ti = (int)((((M < NB) ? M : NB) + 31) >> 5);
		int j = 0, rem = b;
		for(;; ++ j) {
			const int cnt = ((j < ti - 1) ? j : ti - 1) + 1;
			if(rem < cnt)
				break;
			rem -= cnt;
		}
		gemm_tn_staged_tile<32, 32, 16, 16, 0, 0>((int64_t)rem * 32, (int64_t)j * 32, M, N, P, ld, P, ld, C, ld, sm);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if(tid == 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		return;
	}
	if(do_potrf && b == nA) {
		__shared__ int go;
		if(tid == 0) {
			const long long t0 = wall_clock64();
			int ok = 1;
			for(int it = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++ it) { // monotonic counter: its value after this launch's nA producers
				if((it & 15) == 15) {
					if(__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
						ok = 0;
						break;
					}
					if(wall_clock64() - t0 > timeout_ticks) {
						__hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						ok = 0;
						break;
					}
				}
				__builtin_amdgcn_s_sleep(1);
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			go = ok;
		}
		__syncthreads();
		if(!go)
			return;
		// (a fence-free variant -- write-through stores of the sub-tiles, sc1 loads here: potrf_diag_body<true> -- measured
		// the same within noise; the release / acquire pair is the form kept)
		potrf_diag_body<false, 0, 1>(C, ld, n_valid, has_rhs, tinv, info, k0_next, sm);
		return;
	}
	// the rest of the region in 64 x 64 blocks, block columns first
	const int64_t q = b - nA - (do_potrf ? 1 : 0);
	const int64_t nbi = (M + 63) >> 6;
	const int64_t bi = q % nbi, bj = q / nbi;
	if(bi > bj || (bi < 2 && bj < 2))
		return; // strictly below the diagonal, or part of the first tile
	gemm_tn_staged_tile<64, 64, 16, 16, 0, 0>(bi * 64, bj * 64, M, N, P, ld, P, ld, C, ld, sm);
}

} // namespace spp
#include "spp_dense_la.h" // the lookahead schedule: persistent chain kernel + one bulk launch per step
#include "spp_dense_tail.h" // the streamed tail: one workgroup per tile, the factorization passed on 16 rows at a time
namespace spp {

// backward substitution step for block column k (rows/cols k0 .. k0 + NB):
//   x_k = Tinv_k * y_k ; y_i -= R[i, k-block] x_k for all rows i < k0.
// Every workgroup recomputes x_k (128 x 128 GEMV out of L2); workgroup 0 stores it to `xout`
// (a buffer distinct from y: the other workgroups still read y_k), workgroup b > 0 updates 256 rows.
__global__ __launch_bounds__(256)
void trsv_back_kernel(const double *__restrict__ R, int64_t ld, int64_t k0, int nv, const double *__restrict__ tinv,
	double *__restrict__ y, double *__restrict__ xout)
{
	__shared__ double xk[NB];
	__shared__ double part[2][NB];
	const int tid = threadIdx.x;
	{
		// two threads per row, 64 columns each; 8 independent partial sums keep 8 loads in flight
		const int r = tid & (NB - 1), h = tid >> 7;
		double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for(int c8 = 0; c8 < 64; c8 += 8) {
#pragma unroll
			for(int u = 0; u < 8; ++ u) {
				const int c = h * 64 + c8 + u;
				const double t = tinv[r + c * NB], yy = y[k0 + c];
				s8[u] += (c < nv) ? t * yy : 0.0;
			}
		}
		part[h][r] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
	}
	__syncthreads();
	if(tid < NB)
		xk[tid] = part[0][tid] + part[1][tid];
	__syncthreads();
	if(blockIdx.x == 0 && tid < NB)
		xout[k0 + tid] = xk[tid];
	if(blockIdx.x == 0)
		return;
	const int64_t i = (int64_t)(blockIdx.x - 1) * 256 + tid;
	if(i < k0) {
		const double *row = R + i + k0 * ld;
		// real columns only (the padding of the last block holds the rhs column); 16 loads in flight
		double s16[16];
#pragma unroll
		for(int u = 0; u < 16; ++ u)
			s16[u] = 0;
		int c = 0;
		for(; c + 16 <= nv; c += 16) {
#pragma unroll
			for(int u = 0; u < 16; ++ u)
				s16[u] += row[(int64_t)(c + u) * ld] * xk[c + u];
		}
		for(; c < nv; ++ c)
			s16[0] += row[(int64_t)c * ld] * xk[c];
		double s = 0;
#pragma unroll
		for(int u = 0; u < 16; ++ u)
			s += s16[u];
		y[i] -= s;
	}
}


// Backward substitution as ONE launch: one workgroup per block row (128 rows), last block row first in dispatch
// order; the workgroups run the dependent chain x_last -> ... -> x_0 through global memory instead of through 41
// kernel launches. Workgroup b applies x_k to its rows for k = last .. b+1
// as the x_k are published (flag[k] == epoch of this solve), then solves its diagonal block with the
// stored inverse, publishes x_b and exits. Every wait depends only on workgroups with a larger index, which never
// wait on smaller ones: no cycle. The spin is BOUNDED: on timeout the workgroup raises `err`, publishes
// nothing and exits -- the ones below it time out the same way -- so the grid always drains.
__global__ __launch_bounds__(256)
void trsv_back_chain_kernel(const double *__restrict__ R, int64_t ld, int64_t n, int nblk,
	const double *__restrict__ tinv_all, const double *__restrict__ y, double *xout, int *flags, int epoch, int *err)
{
	__shared__ double xs[NB];
	__shared__ double part[2][NB];
	__shared__ int ok;
	// block row b = nblk - 1 - blockIdx.x: the PRODUCERS (large b) are dispatched first, so a workgroup only ever waits
	// for workgroups dispatched before it -- progress does not depend on how many workgroups are resident (this kernel
	// holds a whole CU: more than 256 block rows, or a CU-masked stream, would otherwise park waiters on every CU)
	const int b = nblk - 1 - (int)blockIdx.x, tid = threadIdx.x, r = tid & (NB - 1), h = tid >> 7;
	const int64_t r0 = (int64_t)NB * b;
	// Everything that does not depend on an x_k is fetched ahead of the wait for it: the 64 entries of
	// this thread's half row of the inverse diagonal block (kept for the whole kernel) and of the next
	// R tile (refilled right after use). One workgroup per CU, one wave per SIMD: 512 VGPRs per lane.
	double tv[64], rt[64];
	{
		const double *tinv = tinv_all + (size_t)b * NB * NB + r + (size_t)(64 * h) * NB;
#pragma unroll
		for(int c = 0; c < 64; ++ c)
			tv[c] = tinv[(size_t)c * NB];
	}
	auto load_tile = [&](int k) {
		const double *row = R + (r0 + r) + ((int64_t)NB * k + 64 * h) * ld;
#pragma unroll
		for(int c = 0; c < 64; ++ c)
			rt[c] = row[(int64_t)c * ld];
	};
	if(nblk - 1 > b)
		load_tile(nblk - 1);
	double acc = (h == 0 && r0 + r < n) ? y[r0 + r] : 0.0; // each row is split over two threads (64 columns each)
	for(int k = nblk - 1; k > b; -- k) {
		if(tid == 0) {
			int spins = 0;
			// relaxed polling by ONE thread: an ACQUIRE load at agent scope invalidates the XCD's L2 on
			// every poll (40 pollers thrash the caches of the workgroups that stream R), and 128 threads
			// polling the data itself (x_k preset to a NaN pattern) was 2x slower than flag + fence;
			// x_k is read with agent-scope atomic loads, which go to the coherent level on their own
			while(__hip_atomic_load(&flags[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && spins < (1 << 22)) {
				__builtin_amdgcn_s_sleep(1);
				++ spins;
			}
			ok = spins < (1 << 22);
		}
		__syncthreads();
		if(!ok) {
			if(tid == 0)
				*err = 1;
			return;
		}
		if(tid < NB)
			xs[tid] = __hip_atomic_load(&xout[(int64_t)NB * k + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		const int64_t c0 = (int64_t)NB * k;
		const int nvk = (int)((n - c0 < NB) ? (n - c0) : NB); // real columns of block k (its padding holds the rhs)
		double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for(int c = 0; c < 64; ++ c)
			s8[c & 7] += (64 * h + c < nvk) ? rt[c] * xs[64 * h + c] : 0.0;
		acc -= ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
		if(k - 1 > b)
			load_tile(k - 1); // in flight while this workgroup waits for x_{k-1}
		__syncthreads(); // xs is overwritten by the next block
	}
	part[h][r] = acc;
	__syncthreads();
	if(tid < NB)
		xs[tid] = part[0][tid] + part[1][tid]; // y_b with everything to its right eliminated
	__syncthreads();
	{
		const int nv = (int)((n - r0 < NB) ? (n - r0) : NB);
		double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for(int c = 0; c < 64; ++ c)
			s8[c & 7] += (64 * h + c < nv) ? tv[c] * xs[64 * h + c] : 0.0;
		part[h][r] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
	}
	__syncthreads();
	// hand-off without a fence (MI355X_MICROARCH.md, valid forms: write-through payload on both sides): x_b is stored with
	// agent-scope atomic (sc1) stores and read with agent-scope atomic loads, every storing wave drains its stores
	// (s_waitcnt vmcnt(0)), the workgroup barrier orders them before the one relaxed flag store. The __threadfence() +
	// release store this replaces wrote back and invalidated the L2 once per block row: ~4 us of the 8.5 us per hop.
	if(tid < NB)
		__hip_atomic_store(&xout[r0 + tid], part[0][tid] + part[1][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if(tid == 0)
		__hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// --------------------------------------------------------------------------------------------------
// Backward substitution, second form: the dependent chain x_last -> ... -> x_0 runs inside ONE workgroup.
// In trsv_back_chain_kernel every hop of the chain crosses from one workgroup to the next through memory (payload,
// drain, flag, poll, payload load: ~7 us per block row, 41 of them on Venice). Here
//   workgroup 0 (the chain) computes  x_b = Tinv_b (w_b - R_{b, b+1} x_{b+1})  for b = last .. 0, the tile right of the
//                                     diagonal and the block inverse prefetched into registers one hop ahead, x_{b+1}
//                                     in LDS: nothing on its path waits for another workgroup as long as w_b is there;
//   workgroup 1 + i (helper of block row b = last - i) streams the rest of the row as the x_k arrive,
//                                     w_b = y_b - sum_{k > b + 1} R_{b, k} x_k, and hands w_b over -- it is done two
//                                     hops before the chain needs it.
// What bounds a hop is what ONE workgroup can fetch (tile + triangular inverse = 208 KB at ~65 GB/s: 3.4 us; without the
// fetches a hop takes 1.6 us, and the chain never waits for a helper). Two tiles in the chain were slower (more to
// fetch), four tiles in registers for a deeper prefetch are the CU's whole register file.
// Hand-overs carry no flag and no fence: every element travels as the pair {value, bits(value) ^ K} (K: a 64-bit
// constant of this solve), both words written and read with agent-scope relaxed atomics (write-through / L2-bypassing on
// gfx950). A reader accepts an element when the pair is consistent: a pair from an earlier solve, a pair not yet
// written and a torn pair (one word new, one old) all fail the test (a false accept needs a 64-bit collision) and are
// simply read again. So the producer never drains its stores and the consumer needs no second, dependent load.
// Dispatch order = chain, then helpers from the last block row down: a workgroup only ever waits for data that
// workgroups dispatched before it produce without waiting for a later one (helper b needs x_k, k > b + 1, which need
// helpers > b + 1 only), so progress does not depend on how many workgroups are resident. Spins are bounded (err).
// --------------------------------------------------------------------------------------------------
struct TrsvPay { double v; unsigned long long c; };

__device__ __forceinline__ void pay_store(TrsvPay *p, double v, unsigned long long K)
{
	__hip_atomic_store(&p->v, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&p->c, (unsigned long long)__double_as_longlong(v) ^ K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool pay_load(const TrsvPay *p, double &v, unsigned long long K)
{
	v = __hip_atomic_load(&p->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const unsigned long long c = __hip_atomic_load(&p->c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return ((unsigned long long)__double_as_longlong(v) ^ c) == K;
}

// A 128 x 128 tile of a column-major matrix in the registers of 4 waves: lane l holds rows 2 l, 2 l + 1 (one 16-byte
// load), wave w the columns w, w + 4, ..: one wave load = one whole tile column, 1 KB contiguous. Measured for one
// workgroup pulling tiles out of the factor (tools/tile_load_bw.hip): 100 GB/s this way, 75 GB/s with 8-byte loads of 64
// consecutive rows, 38 GB/s with two threads per row -- and the chain below lives on what ONE workgroup can pull.
template <int NW>
struct TileRegs { double2 v[NB / NW]; };

// TRI: the tile is upper triangular (a block inverse): a lane whose rows lie below column c does not fetch it
template <bool TRI, int NW>
__device__ __forceinline__ void tile_fetch(TileRegs<NW> &t, const double *T, const int64_t ld, const int lane, const int wave)
{
#pragma unroll
	for(int j = 0; j < NB / NW; ++ j) {
		const int c = wave + NW * j;
		// (address = wave-uniform column base in scalar registers + one unsigned 32-bit lane offset)
		const double *col = T + (int64_t)c * ld;
		const uint32_t loff = 2u * (uint32_t)lane;
		if(!TRI || 2 * lane <= c)
			t.v[j] = *(const double2*)(col + loff);
		else
			t.v[j] = make_double2(0, 0);
	}
}

// a += tile . x over this wave's columns (columns >= nv masked: the padding of the last block holds the rhs column)
template <int NW>
__device__ __forceinline__ void tile_dot(double2 &a, const TileRegs<NW> &t, const double *x, const int wave, const int nv)
{
	double2 s0 = make_double2(0, 0), s1 = make_double2(0, 0);
	if(nv >= NB) {
#pragma unroll
		for(int j = 0; j < NB / NW; j += 2) {
			const double x0 = x[wave + NW * j], x1 = x[wave + NW * j + NW];
			s0.x += t.v[j].x * x0; s0.y += t.v[j].y * x0;
			s1.x += t.v[j + 1].x * x1; s1.y += t.v[j + 1].y * x1;
		}
	} else {
#pragma unroll
		for(int j = 0; j < NB / NW; ++ j) {
			const int c = wave + NW * j;
			const double xv = (c < nv) ? x[c] : 0.0; // (wave-uniform)
			s0.x += (c < nv) ? t.v[j].x * xv : 0.0;
			s0.y += (c < nv) ? t.v[j].y * xv : 0.0;
		}
	}
	a.x += s0.x + s1.x;
	a.y += s0.y + s1.y;
}

// NW waves (8: measured 140 us on the Venice system, 4 waves 194 us, 16 waves 142 us)
// MFORM: the chain applies ONE precomputed tile per hop, x_b = u_b - M_b x_{b+1} with M_b = Tinv_b R_{b, b+1}
// (trsv_m_kernel) and u_b = Tinv_b w_b formed by the helper -- 128 KB to fetch per hop instead of 208, one product and two
// barriers instead of two and four, and two hops of prefetch fit the registers.
template <int NW, bool MFORM>
__global__ __launch_bounds__(64 * NW)
void trsv_back_chain2_kernel(const double *__restrict__ R, int64_t ld, int64_t n, int nblk,
	const double *__restrict__ tinv_all, const double *__restrict__ mbuf, double *y, TrsvPay *xpay, TrsvPay *wpay, unsigned long long K, int *err)
{
	__shared__ double xs[2][NB];   // chain: x_b in xs[b & 1] ; helper: xs[0 / 1] = the x_k being applied
	__shared__ double zs[NB];
	__shared__ __attribute__((aligned(16))) double part[NW][NB]; // the waves' partial sums per row
	__shared__ int ok[2];
	constexpr int L = 1; // tiles right of the diagonal the chain applies itself (two measured slower: the chain is bound by what it fetches)
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (wave: a scalar -- column bases stay in scalar registers)
	constexpr int SPIN_MAX = 1 << 22;
	auto part_sum = [&](const int row) {
		double s = 0;
#pragma unroll
		for(int w = 0; w < NW; w += 2)
			s += part[w][row] + part[w + 1][row];
		return s;
	};
	// (lds_barrier, not __syncthreads: the latter waits for every outstanding global load -- the prefetched tiles)
	if(blockIdx.x > 0) {
		// ---- helper of block row b
		const int b = nblk - (int)blockIdx.x;
		const int64_t r0 = (int64_t)NB * b;
		TileRegs<NW> ra, rb; // the tile being applied and the one after it (static slots: no indexed register arrays)
		TileRegs<NW> tvh;    // MFORM: this block row's inverse diagonal block
		if(MFORM)
			tile_fetch<true, NW>(tvh, tinv_all + (size_t)b * NB * NB, NB, lane, wave);
		const int klast = b + L + 1; // tiles k = nblk - 1 .. klast
		double2 acc = make_double2(0, 0);
		const double yb = (tid < NB && r0 + tid < n) ? y[r0 + tid] : 0.0; // (fetched now: not behind the last x_k)
		bool alive = true;
		// one tile: wait for x_k (slot: which half of xs), apply, refill the register slot with tile k - 2
		auto step = [&](const int k, TileRegs<NW> &t, const int slot) {
			double *xk = xs[slot];
			if(tid < 64) { // one wave polls the payload of x_k: two elements per lane
				const TrsvPay *src = xpay + (size_t)k * NB + 2 * tid;
				double v0 = 0, v1 = 0;
				int spins = 0;
				for(;;) {
					const bool g0 = pay_load(src, v0, K), g1 = pay_load(src + 1, v1, K);
					if(__all(g0 && g1) || ++ spins >= SPIN_MAX)
						break;
					__builtin_amdgcn_s_sleep(2);
				}
				xk[2 * tid] = v0;
				xk[2 * tid + 1] = v1;
				if(tid == 0)
					ok[slot] = spins < SPIN_MAX;
			}
			lds_barrier();
			if(!ok[slot]) {
				if(tid == 0)
					*err = 1;
				alive = false;
				return;
			}
			const int64_t c0 = (int64_t)NB * k;
			const int nvk = (int)((n - c0 < NB) ? (n - c0) : NB); // real columns of block k (its padding holds the rhs)
			double2 s = make_double2(0, 0);
			tile_dot<NW>(s, t, xk, wave, nvk);
			acc.x -= s.x;
			acc.y -= s.y;
			asm volatile("" ::: "memory"); // (the refill reuses the registers just released)
			__builtin_amdgcn_sched_barrier(0);
			if(k - 2 >= klast)
				tile_fetch<false, NW>(t, R + r0 + (int64_t)NB * (k - 2) * ld, ld, lane, wave); // in flight while this workgroup waits for x_{k-1}
			// (no second barrier: x_{k-1} goes to the other half of xs, and the barrier of that round orders the write of
			// x_{k-2} behind this round's reads)
		};
		int k = nblk - 1;
		if(k >= klast)
			tile_fetch<false, NW>(ra, R + r0 + (int64_t)NB * k * ld, ld, lane, wave);
		if(k - 1 >= klast)
			tile_fetch<false, NW>(rb, R + r0 + (int64_t)NB * (k - 1) * ld, ld, lane, wave);
		while(k >= klast && alive) {
			step(k, ra, 0);
			-- k;
			if(k < klast || !alive)
				break;
			step(k, rb, 1);
			-- k;
		}
		if(!alive)
			return;
		*(double2*)&part[wave][2 * lane] = acc;
		lds_barrier();
		if(!MFORM) {
			if(tid < NB)
				pay_store(wpay + (size_t)b * NB + tid, yb + part_sum(tid), K);
			return;
		}
		// u_b = Tinv_b w_b
		if(tid < NB)
			zs[tid] = yb + part_sum(tid);
		lds_barrier();
		{
			double2 s2 = make_double2(0, 0);
			tile_dot<NW>(s2, tvh, zs, wave, (int)((n - r0 < NB) ? (n - r0) : NB));
			*(double2*)&part[wave][2 * lane] = s2;
		}
		lds_barrier();
		if(tid < NB)
			pay_store(wpay + (size_t)b * NB + tid, part_sum(tid), K);
		return;
	}
	if(MFORM) {
		// ---- the chain, M form: two register slots for the tiles M_b, fetched two hops ahead
		TileRegs<NW> ma, mb2;
		double uv = 0;
		unsigned long long uc = 0;
		auto ask_u = [&](int b, double &v, unsigned long long &c) {
			const TrsvPay *p = wpay + (size_t)b * NB + tid;
			v = __hip_atomic_load(&p->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			c = __hip_atomic_load(&p->c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		};
		auto fetch_m = [&](int b, TileRegs<NW> &t) {
			if(b >= 0 && b + 1 < nblk)
				tile_fetch<false, NW>(t, mbuf + (size_t)b * NB * NB, NB, lane, wave);
		};
		auto hop = [&](const int b, TileRegs<NW> &t) {
			const int64_t r0 = (int64_t)NB * b;
			{
				double2 s2 = make_double2(0, 0);
				if(b + 1 < nblk)
					tile_dot<NW>(s2, t, xs[(b + 1) & 1], wave, NB); // (the columns of M_b beyond the next block's pivots are zero)
				*(double2*)&part[wave][2 * lane] = s2;
			}
			asm volatile("" ::: "memory");
			__builtin_amdgcn_sched_barrier(0);
			fetch_m(b - 2, t);
			lds_barrier();
			if(tid < NB) {
				int spins = 0;
				while(((unsigned long long)__double_as_longlong(uv) ^ uc) != K && spins < SPIN_MAX) {
					__builtin_amdgcn_s_sleep(1);
					ask_u(b, uv, uc);
					++ spins;
				}
				if(spins >= SPIN_MAX)
					*err = 1;
				const double xr = uv - part_sum(tid);
				xs[b & 1][tid] = xr;
				pay_store(xpay + (size_t)b * NB + tid, xr, K);
				if(r0 + tid < n)
					y[r0 + tid] = xr;
				if(b > 0)
					ask_u(b - 1, uv, uc);
			}
			lds_barrier();
		};
		fetch_m(nblk - 2, mb2);
		if(tid < NB)
			ask_u(nblk - 1, uv, uc);
		// hop b uses slot ((nblk - 1 - b) & 1): the last block row has no tile; then mb2, ma, mb2, ..
		for(int b = nblk - 1; b >= 0; b -= 2) {
			hop(b, ma);      // (b = nblk - 1: no product; refills ma with the tile of hop b - 2)
			if(b >= 1)
				hop(b - 1, mb2);
		}
		return;
	}
	// ---- the chain. Threads 0 .. 127 own row tid of the current block for the hand-overs (w in, x out).
	// Tile (b, b+1) and the block inverse of hop b are fetched one hop ahead, each right after its use. (Two hops ahead
	// would need four tiles in registers = the whole register file of the CU: what one workgroup can pull, 100 GB/s and
	// latency-bound, is what bounds a hop -- 208 KB in ~3.4 us.)
	TileRegs<NW> t1a, tva;
	auto ask_w = [&](int b, double &v, unsigned long long &c) { // both words of w_b[tid], used one hop later
		const TrsvPay *p = wpay + (size_t)b * NB + tid;
		v = __hip_atomic_load(&p->v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		c = __hip_atomic_load(&p->c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	auto fetch_tile = [&](int b, TileRegs<NW> &t1) { // tile (b, b+1)
		if(b >= 0 && b + 1 < nblk)
			tile_fetch<false, NW>(t1, R + (int64_t)NB * b + (int64_t)NB * (b + 1) * ld, ld, lane, wave);
	};
	auto fetch_inv = [&](int b, TileRegs<NW> &tv) {
		if(b >= 0)
			tile_fetch<true, NW>(tv, tinv_all + (size_t)b * NB * NB, NB, lane, wave);
	};
	double wv = 0;
	unsigned long long wc = 0;
	auto hop = [&](const int b, TileRegs<NW> &t1, TileRegs<NW> &tv) {
		const int64_t r0 = (int64_t)NB * b;
		// A: this wave's share of R_{b, b+1} x_{b+1}
		{
			double2 s = make_double2(0, 0);
			if(b + 1 < nblk) {
				const int64_t c0 = (int64_t)NB * (b + 1);
				tile_dot<NW>(s, t1, xs[(b + 1) & 1], wave, (int)((n - c0 < NB) ? (n - c0) : NB));
			}
			*(double2*)&part[wave][2 * lane] = s;
		}
		// (compiler fence: the prefetch must reuse the registers the products above just released -- hoisted above them
		// it would double the register need)
		asm volatile("" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		fetch_tile(b - 1, t1);
		lds_barrier();
		// B: z = w_b - sum. w_b was asked for one hop ago; read again until the pair is consistent (the helper is
		// normally far ahead)
		if(tid < NB) {
			int spins = 0;
			while(((unsigned long long)__double_as_longlong(wv) ^ wc) != K && spins < SPIN_MAX) {
				__builtin_amdgcn_s_sleep(1);
				ask_w(b, wv, wc);
				++ spins;
			}
			if(spins >= SPIN_MAX)
				*err = 1; // (the hop goes on with garbage: every workgroup still drains)
			zs[tid] = wv - part_sum(tid);
		}
		lds_barrier();
		// C: this wave's share of Tinv_b z
		{
			double2 s = make_double2(0, 0);
			tile_dot<NW>(s, tv, zs, wave, (int)((n - r0 < NB) ? (n - r0) : NB));
			*(double2*)&part[wave][2 * lane] = s;
		}
		asm volatile("" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		fetch_inv(b - 1, tv);
		lds_barrier();
		// D: x_b out
		if(tid < NB) {
			const double xr = part_sum(tid);
			xs[b & 1][tid] = xr;
			pay_store(xpay + (size_t)b * NB + tid, xr, K);
			if(r0 + tid < n)
				y[r0 + tid] = xr; // in place: the helper of this block row read y_b before it handed over w_b
			if(b > 0)
				ask_w(b - 1, wv, wc); // (two hops ahead measured slower: the wait for w_b then also covers the younger request)
		}
		lds_barrier();
	};
	fetch_inv(nblk - 1, tva);
	if(tid < NB)
		ask_w(nblk - 1, wv, wc);
	for(int b = nblk - 1; b >= 0; -- b)
		hop(b, t1a, tva);
}

// M_b = Tinv_b R_{b, b+1} for the M form of the backward substitution: workgroup (b, s) forms the 16 columns 16 s .. of M_b,
// one 16 x 16 tile per wave (8 waves), Tinv_b's upper triangle and the 128 x 16 slab of R staged in LDS. The columns of the
// last block beyond its pivots (identity padding / the right-hand side column) come out as zeros.
constexpr int TRSVM_LDS_DOUBLES = NB * TS + 16 * TS;
__global__ __launch_bounds__(1024)
void trsv_m_kernel(const double *__restrict__ R, int64_t ld, int64_t n, const double *__restrict__ tinv_all, double *__restrict__ mbuf)
{
	extern __shared__ double sm[];
	double *Ti = sm, *Xs = sm + NB * TS; // Tinv_b[i][k] at i + k * TS ; X[k][j] at k + j * TS (16 columns)
	const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
	const double *tv = tinv_all + (size_t)b * NB * NB;
	// 16-byte pieces of the upper tile triangle (rows 2 p, 2 p + 1 of column k)
	for(int e = tid; e < NB * NB / 2; e += 1024) {
		const int i = (e & (NB / 2 - 1)) * 2, k = e >> 6;
		if((i >> 4) <= (k >> 4)) {
			const double2 v = *(const double2*)(tv + i + (size_t)k * NB);
			Ti[i + k * TS] = (i <= k) ? v.x : 0.0; // (only the upper triangle of a block inverse is defined)
			Ti[i + 1 + k * TS] = (i + 1 <= k) ? v.y : 0.0;
		}
	}
	const int64_t c0 = (int64_t)NB * (b + 1) + 16 * sl;
	for(int e = tid; e < NB * 16; e += 1024) {
		const int k = e & (NB - 1), j = e >> 7;
		Xs[k + j * TS] = (c0 + j < n) ? R[(int64_t)NB * b + k + (c0 + j) * ld] : 0.0;
	}
	__syncthreads();
	// tile (it, sl) by two waves: the even and the odd k tiles of it .. 7
	const int it = wave & 7, half = wave >> 3;
	v4f64 acc = (v4f64){0, 0, 0, 0};
	for(int kt = it + half; kt < NB / 16; kt += 2) {
		const v4f64 m = tile_atb(Ti + 16 * it + 16 * kt * TS, TS, 1, Xs + 16 * kt, 1, TS, lane); // sum_k Tinv[16 it + i][16 kt + k] X[16 kt + k][j]
		acc += m;
	}
	// (the odd half's partial tile goes through a tile of the image's unused lower triangle: (7, it), it < 7; tile row 7
	// has no odd half)
	double *Pp = Ti + 16 * 7 + (16 * it) * TS;
	if(half && it < 7) {
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			Pp[(l4 + 4 * r) + l15 * TS] = acc[r];
	}
	__syncthreads();
	if(!half) {
		double *M = mbuf + (size_t)b * NB * NB;
#pragma unroll
		for(int r = 0; r < 4; ++ r)
			M[(16 * it + l4 + 4 * r) + (size_t)(16 * sl + l15) * NB] = acc[r] + (it < 7 ? Pp[(l4 + 4 * r) + l15 * TS] : 0.0);
	}
}

__global__ void set_info_kernel(int *info) { info[0] = 0; info[1] = 0; }

// padding diagonal = 1 (rows/cols >= n); the rows >= n of the rhs column (column n) are cleared so
// that the solves never touch non-finite garbage
__global__ void pad_diag_kernel(double *S, int64_t ld, int64_t n)
{
	const int64_t i = n + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i < ld) {
		S[i + n * ld] = 0.0;
		S[i + i * ld] = 1.0;
	}
}

void dense_set_padding(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n)
{
	hipLaunchKernelGGL(pad_diag_kernel, dim3((unsigned)((ld - n + 255) / 256)), dim3(256), 0, ctx->stream,
		d_A, ld, n);
}

// --------------------------------------------------------------------------------------------------
// host drivers
// --------------------------------------------------------------------------------------------------
// info[0]: first failing pivot + 1, info[1]: timeout of the backward-substitution chain, info[2]: abort flag of
// the device-flag hand-offs (stays set until the host has seen it), info[3]: abort flag of the sparse path's
// dependency-driven launches -- zeroed once, when the buffer is created
static void ensure_info(spp_ctx *ctx)
{
	if(!ctx->dense.info.p) {
		ctx->dense.info.reserve(4);
		SPP_HIP_CHECK(hipMemsetAsync(ctx->dense.info.p, 0, 4 * sizeof(int), ctx->stream));
	}
}

// A stream with a CU mask takes ~10 ms to create (measured: a fifth of the whole analysis of a Venice-sized problem): a
// closed context parks its bulk stream here and the next context on the same device takes it over. (The mask depends on
// SPP_AUX_RESERVE_CUS alone, which is read once per process.)
static std::mutex aux_pool_mutex;
static std::vector<std::pair<int, hipStream_t> > aux_pool;

static hipStream_t dense_aux_take(int device)
{
	std::lock_guard<std::mutex> lock(aux_pool_mutex);
	for(size_t i = 0; i < aux_pool.size(); ++ i)
		if(aux_pool[i].first == device) {
			hipStream_t s = aux_pool[i].second;
			aux_pool.erase(aux_pool.begin() + i);
			return s;
		}
	return nullptr;
}

void dense_aux_park(int device, hipStream_t s)
{
	if(hipStreamSynchronize(s) != hipSuccess) {
		(void)hipGetLastError();
		(void)hipStreamDestroy(s);
		return;
	}
	std::lock_guard<std::mutex> lock(aux_pool_mutex);
	if(aux_pool.size() >= 16) {
		(void)hipStreamDestroy(s);
		return;
	}
	// No stream may be left alive at process exit (the runtime's own teardown of a leftover CU-masked stream crashed under
	// rocprofv3): the first parked stream registers a handler, which runs before the destructors of everything loaded earlier
	static bool registered = false;
	if(!registered) {
		registered = true;
		atexit([]() {
			std::lock_guard<std::mutex> lock2(aux_pool_mutex);
			for(size_t i = 0; i < aux_pool.size(); ++ i)
				(void)hipStreamDestroy(aux_pool[i].second);
			aux_pool.clear();
		});
	}
	aux_pool.push_back(std::make_pair(device, s));
}

static void ensure_dense_work(spp_ctx *ctx, int64_t nblk)
{
	VClock clk("dense workspaces");
	ensure_info(ctx);
	clk.lap("status words");
	if(ctx->dense.tinv_all.cap < (size_t)nblk * NB * NB) {
		// zeroed once: the lookahead schedule never stores the zeros below the diagonal of a block's inverse
		ctx->dense.tinv_all.reserve((size_t)nblk * NB * NB);
		SPP_HIP_CHECK(hipMemsetAsync(ctx->dense.tinv_all.p, 0, ctx->dense.tinv_all.cap * sizeof(double), ctx->stream));
	}
	ctx->dense.xtmp.reserve((size_t)(nblk + 1) * NB);
	clk.lap("block inverses");
	if(!ctx->dense.aux && (ctx->dense.aux = dense_aux_take(ctx->device)) != nullptr) {
		SPP_HIP_CHECK(hipEventCreateWithFlags(&ctx->dense.ev[0], hipEventDisableTiming));
		SPP_HIP_CHECK(hipEventCreateWithFlags(&ctx->dense.ev[1], hipEventDisableTiming));
	}
	if(!ctx->dense.aux) {
		// The bulk stream is created with a CU mask that leaves the first n CUs (SPP_AUX_RESERVE_CUS, default 32)
		// to the chain: under a running bulk update every wave slot of the chip is taken, potrf_diag and the
		// tile-row workgroups otherwise wait for bulk workgroups to retire (measured: potrf_diag 33 -> 50..100 us,
		// tile row 10 -> 55 us). 8 CUs already help, 32..64 are best; beyond that the bulk update itself suffers.
		const char *r = getenv("SPP_AUX_RESERVE_CUS");
		const int reserve = r ? atoi(r) : 32;
		if(reserve > 0) {
			int ncu = 0; // (one attribute: hipGetDeviceProperties fills a 1.5 KB record through dozens of driver queries, milliseconds)
			SPP_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
			const int nw = (ncu + 31) / 32;
			std::vector<uint32_t> mask((size_t)nw, 0u);
			for(int c = reserve; c < ncu; ++ c)
				mask[(size_t)c / 32] |= 1u << (c % 32);
			if(hipExtStreamCreateWithCUMask(&ctx->dense.aux, (uint32_t)nw, mask.data()) != hipSuccess) {
				(void)hipGetLastError(); // no CU masking on this stack: plain low-priority stream
				ctx->dense.aux = nullptr;
			}
		}
		if(!ctx->dense.aux) {
			// without a mask: at least the LOWEST priority, so that workgroups of the serial chain (on ctx->stream) are
			// dispatched first whenever a CU frees up under a running trailing update
			int prio_lo = 0, prio_hi = 0;
			SPP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
			SPP_HIP_CHECK(hipStreamCreateWithPriority(&ctx->dense.aux, hipStreamNonBlocking, prio_lo));
		}
		SPP_HIP_CHECK(hipEventCreateWithFlags(&ctx->dense.ev[0], hipEventDisableTiming));
		SPP_HIP_CHECK(hipEventCreateWithFlags(&ctx->dense.ev[1], hipEventDisableTiming));
	}
	clk.lap("bulk stream + events");
	static uint64_t attr_set_seen = 0;
	if(first_on_this_device(attr_set_seen)) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)potrf_diag_kernel,
			hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_DOUBLES * (int)sizeof(double)));
	}
}

// Factor the n_pad x n_pad matrix in d_A (ld = n_pad multiple of 128, n < n_pad real columns; column
// n is carried along as right-hand side). Returns SPP_OK / SPP_NOT_POSDEF.
//
// Schedule (lookahead over two streams): after the row panel k is solved, the trailing update is
// split into (1) the tile row k+1 -- all the next diagonal block and the next row panel need -- and
// (3) the rest. The serial chain potrf_diag(k+1) + trsm(k+1) runs on the auxiliary stream while (3)
// keeps the whole chip busy on the main stream.
// Generic driver: eliminates `nsteps` diagonal blocks of the matrix in d_A.
//   n      number of pivot scalars (the last block may be partial: n_valid = n - k0 < 128, padding rows
//          must then be exact identity and, if has_rhs, column n carries a right-hand side)
//   rows   extent of the trailing region that receives updates (n for a full factorization, the
//          padded front height for a partial front factorization)
//   ncols  columns carried along (rows + 1 with a rhs column)
// Does not synchronize; failures are recorded in ctx->dense.info (first failing pivot + 1).
static void dense_factor_steps_enqueue(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n, int64_t rows, int64_t ncols,
	int64_t nsteps, bool has_rhs, bool use_flags = false, bool allow_fused = true);
static bool flag_schedule_usable(spp_ctx *ctx);
static void flag_signal(hipStream_t st, int *flag, int value);
static void flag_wait(spp_ctx *ctx, hipStream_t st, const int *flag, int value, double timeout_ms = 500.0);

// ---- lookahead schedule (spp_dense_la.h): host side -------------------------------------------------------------------
// Streams: the chain kernel runs on a stream of its own that is CU-masked to the `reserve` CUs the bulk stream's mask
// leaves alone (it IS the reservation: 1 + LA_G1 + g2 <= reserve workgroups, one per CU, resident for the whole
// factorization); the bulk launches go to the masked bulk stream. Both are forked from / joined into the ctx stream by
// events, once per factorization.
static void la_setup_streams(spp_ctx *ctx)
{
	DenseWork &dw = ctx->dense;
	if(dw.la_state != 0)
		return;
	dw.la_state = -1;
	const char *e = getenv("SPP_DENSE_LA"); // 1: the lookahead schedule (spp_dense_la.h); default: the two-stream schedule of rounds 1-2
	if(!e || !atoi(e))
		return;
	hipDeviceProp_t prop;
	SPP_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
	const int ncu = prop.multiProcessorCount;
	const char *r = getenv("SPP_AUX_RESERVE_CUS");
	const int reserve = r ? atoi(r) : 32;
	if(ncu < 128 || reserve < 1 + LA_G1 + 1 || reserve > ncu / 2 || !dw.aux)
		return; // a small partition: the chain kernel's workgroups + a useful bulk side do not fit
	dw.la_ncu = ncu;
	dw.la_reserve = reserve;
	const char *m = getenv("SPP_LA_CHAIN_MASK"); // 0: the chain kernel on an unmasked stream
	const int nw = (ncu + 31) / 32;
	std::vector<uint32_t> mask((size_t)nw, 0u);
	for(int c = 0; c < reserve; ++ c)
		mask[(size_t)c / 32] |= 1u << (c % 32);
	if((m && !atoi(m)) || hipExtStreamCreateWithCUMask(&dw.chain, (uint32_t)nw, mask.data()) != hipSuccess) {
		(void)hipGetLastError();
		SPP_HIP_CHECK(hipStreamCreateWithFlags(&dw.chain, hipStreamNonBlocking));
	}
	SPP_HIP_CHECK(hipEventCreateWithFlags(&dw.ev_chain, hipEventDisableTiming));
	// the chain stream and the bulk stream must run CONCURRENTLY (two streams that share a hardware queue would make the
	// bulk launches wait for the end of the chain kernel, which waits for them): wait enqueued first, both directions
	dw.sync.reserve(16);
	SPP_HIP_CHECK(hipMemsetAsync(dw.sync.p, 0, 16 * sizeof(int), ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	hipStream_t st[2] = {dw.chain, dw.aux};
	for(int p = 0; p < 2; ++ p) {
		flag_wait(ctx, st[p], dw.sync.p + 8 + p, 1, 20.0);
		flag_signal(st[1 - p], dw.sync.p + 8 + p, 1);
	}
	for(int i = 0; i < 2; ++ i)
		SPP_HIP_CHECK(hipStreamSynchronize(st[i]));
	int h_abort = 0;
	SPP_HIP_CHECK(hipMemcpy(&h_abort, dw.info.p + 2, sizeof(int), hipMemcpyDeviceToHost));
	if(h_abort) {
		SPP_HIP_CHECK(hipMemset(dw.info.p + 2, 0, sizeof(int)));
		return;
	}
	static uint64_t attr_seen = 0;
	if(first_on_this_device(attr_seen)) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)la_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			LA_LDS_DOUBLES * (int)sizeof(double)));
	}
	dw.la_state = 1;
}

static bool la_usable(spp_ctx *ctx, int64_t nsteps)
{
	DenseWork &dw = ctx->dense;
	if(nsteps < 6 || dw.sync_state < 0)
		return false;
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if(hipStreamIsCapturing(ctx->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
		return false; // counters are zeroed and polled inside one call: not for a captured graph
	la_setup_streams(ctx);
	return dw.la_state == 1;
}

static void dense_factor_lookahead(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n, int64_t rows, int64_t ncols,
	int64_t nsteps, bool has_rhs)
{
	DenseWork &dw = ctx->dense;
	hipStream_t s = ctx->stream, sb = dw.aux, sc = dw.chain;
	LaArgs a;
	a.A = d_A;
	a.ld = ld;
	a.n_id = (dw.ident_from >= 0 && dw.ident_from < n) ? dw.ident_from : n;
	a.rows = rows;
	a.ncols = ncols;
	a.nsteps = (int)nsteps;
	a.nt = (int)((ncols + NB - 1) / NB);
	a.has_rhs = has_rhs ? 1 : 0;
	a.g1 = LA_G1;
	static int g2_env = -1, acq = -1, trace_n = -1;
	static int64_t slots_env = -1;
	if(g2_env < 0) {
		const char *e = getenv("SPP_LA_G2");
		g2_env = e ? atoi(e) : 0;
		e = getenv("SPP_LA_BULK_ACQUIRE");
		acq = e ? atoi(e) : 1;
		e = getenv("SPP_LA_SLOTS");
		slots_env = e ? atol(e) : 0;
		e = getenv("SPP_LA_TRACE"); // n: the n-th factorization prints when its diagonal blocks started / ended
		trace_n = e ? atoi(e) : 0;
	}
	a.g2 = g2_env > 0 ? std::min(g2_env, dw.la_reserve - 1 - LA_G1) : dw.la_reserve - 1 - LA_G1;
	a.tinv_all = dw.tinv_all.p;
	a.info = dw.info.p;
	const size_t nints = (la_counter_ints(a.nsteps, a.nt) + 3) & ~(size_t)3;
	dw.la_cnt.reserve(nints);
	a.cnt = dw.la_cnt.p;
	a.abort = dw.info.p + 2;
	a.timeout_ticks = (long long)(500.0 * 1e5);
	a.bulk_acquire = acq;
	{
		static int u = -1;
		if(u < 0) {
			const char *e = getenv("SPP_TILE_444");
			u = e ? atoi(e) : 0;
		}
		static uint64_t bulk_attr_seen = 0;
		if(first_on_this_device(bulk_attr_seen))
			SPP_HIP_CHECK(hipFuncSetAttribute((const void*)la_bulk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
				LA_BULK_LDS_DOUBLES * (int)sizeof(double)));
		a.use444 = u;
	}
	a.trace = nullptr;
	static int la_calls = 0;
	const bool tracing = trace_n > 0 && ++ la_calls == trace_n;
	if(tracing) {
		const size_t nw = (size_t)12 * nsteps + 16;
		dw.la_trace.reserve(nw);
		std::vector<long long> init(nw, 0);
		for(int64_t k = 0; k < nsteps; ++ k)
			init[8 * nsteps + 8 + 4 * k] = (long long)((~0ull) >> 1); // atomicMin slot
		SPP_HIP_CHECK(hipMemcpyAsync(dw.la_trace.p, init.data(), nw * sizeof(long long), hipMemcpyHostToDevice, s));
		SPP_HIP_CHECK(hipStreamSynchronize(s));
		a.trace = dw.la_trace.p;
	}
	SPP_HIP_CHECK(hipMemsetAsync(a.cnt, 0, nints * sizeof(int), s));
	SPP_HIP_CHECK(hipEventRecord(dw.ev[0], s));
	SPP_HIP_CHECK(hipStreamWaitEvent(sc, dw.ev[0], 0));
	SPP_HIP_CHECK(hipStreamWaitEvent(sb, dw.ev[0], 0));
	hipLaunchKernelGGL(la_chain_kernel, dim3((unsigned)(1 + LA_G1 + a.g2)), dim3(LA_THREADS), LA_LDS_DOUBLES * sizeof(double), sc, a);
	const int64_t slots = slots_env > 0 ? slots_env : 512; // the first multiple of this many tiles of a launch go whole, the rest in quarters
	for(int k = 0; k < a.nsteps; ++ k) {
		const int64_t c1 = (int64_t)NB * (k + 1);
		if(c1 >= ncols || rows - c1 <= 0)
			break;
		const int nr = (int)((rows - c1 + NB - 1) / NB), nc = (int)((ncols - c1 + NB - 1) / NB);
		const int npri = (nc > 2) ? (nc - 2) + std::min(2, nr - 1) : 0; // row 0 from column 2 on, (1, 2), (2, 2)
		const int64_t ntr = nr - 1, ntc = nc - 1;
		const int64_t T = ntr > 0 ? ntr * (ntr + 1) / 2 + (ntc - ntr) * ntr : 0;
		const int64_t n128 = (T / slots) * slots;
		const int64_t grid = 4 * (int64_t)npri + n128 + 4 * (T - n128);
		if(grid > 0)
			hipLaunchKernelGGL(la_bulk_kernel, dim3((unsigned)grid), dim3(1024), LA_BULK_LDS_DOUBLES * sizeof(double), sb,
				a, k, nr, nc, npri, n128);
	}
	SPP_HIP_CHECK(hipEventRecord(dw.ev[1], sb));
	SPP_HIP_CHECK(hipEventRecord(dw.ev_chain, sc));
	SPP_HIP_CHECK(hipStreamWaitEvent(s, dw.ev[1], 0));
	SPP_HIP_CHECK(hipStreamWaitEvent(s, dw.ev_chain, 0));
	SPP_HIP_CHECK(hipGetLastError());
	if(tracing) {
		std::vector<long long> h((size_t)12 * nsteps + 16);
		SPP_HIP_CHECK(hipMemcpyAsync(h.data(), dw.la_trace.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
		SPP_HIP_CHECK(hipStreamSynchronize(s));
		fprintf(stderr, "[spp] lookahead factorization, %d steps: potrf start / duration / gap to the next start (us)\n", a.nsteps);
		for(int k = 0; k < a.nsteps; ++ k)
			fprintf(stderr, "  k %2d  start %8.1f  potrf %5.1f  publish %4.1f | b1 seen +%4.1f  done +%4.1f | c1 seen +%4.1f  done +%4.1f | step %5.1f\n", k,
				(h[8 * k] - h[0]) * 0.01, (h[8 * k + 2] - h[8 * k]) * 0.01, (h[8 * k + 1] - h[8 * k + 2]) * 0.01,
				(h[8 * k + 3] - h[8 * k + 1]) * 0.01, (h[8 * k + 4] - h[8 * k + 1]) * 0.01, (h[8 * k + 5] - h[8 * k + 1]) * 0.01,
				(h[8 * k + 6] - h[8 * k + 1]) * 0.01, k + 1 < a.nsteps ? (h[8 * (k + 1)] - h[8 * k]) * 0.01 : 0.0);
		fprintf(stderr, "[spp] per step: panel (rest) done / bulk launch: first tile starts, priority tiles done, last tile done, busy workgroup-us (all relative to potrf(0) start)\n");
		for(int k = 0; k < a.nsteps; ++ k) {
			const long long *b = h.data() + 8 * nsteps + 8 + 4 * k;
			fprintf(stderr, "  k %2d  panel done %8.1f | bulk start %8.1f  pri done %8.1f  end %8.1f  busy %9.1f\n", k, h[8 * k + 7] ? (h[8 * k + 7] - h[0]) * 0.01 : 0.0,
				b[1] ? (b[0] - h[0]) * 0.01 : 0.0, b[3] ? (b[3] - h[0]) * 0.01 : 0.0, b[1] ? (b[1] - h[0]) * 0.01 : 0.0, b[2] * 0.01);
		}
	}
}

// Which schedule: the lookahead schedule (SPP_DENSE_LA=1), the two-stream schedule with device flags + the fused chain
// kernel (default), or -- inside a stream capture, where monotonic counters and epochs passed as kernel arguments would
// be replayed stale -- the two-stream schedule with events only and the three chain kernels as separate launches.
void dense_factor_steps(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n, int64_t rows, int64_t ncols,
	int64_t nsteps, bool has_rhs)
{
	ensure_dense_work(ctx, nsteps);
	ctx->dense.tail_rows_last = 0;
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	const bool capturing = hipStreamIsCapturing(ctx->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
	if(!capturing && la_usable(ctx, nsteps)) { // lookahead: persistent chain kernel + one bulk launch per step
		dense_factor_lookahead(ctx, d_A, ld, n, rows, ncols, nsteps, has_rhs);
		ctx->dense.tinv_half = 0; // (its chain stores whole inverses)
		return;
	}
	ctx->dense.tinv_half = nsteps;
	const bool flags = !capturing && nsteps >= 4 && flag_schedule_usable(ctx); // cross-stream hand-offs through device flags instead of events
	dense_factor_steps_enqueue(ctx, d_A, ld, n, rows, ncols, nsteps, has_rhs, flags, !capturing);
}

static void flag_signal(hipStream_t st, int *flag, int value)
{
	hipLaunchKernelGGL(flag_signal_kernel, dim3(1), dim3(64), 0, st, flag, value);
}

static FlagWait make_flag_wait(spp_ctx *ctx, const int *flag, int value, double timeout_ms = 500.0)
{
	return FlagWait{flag, value, ctx->dense.info.p + 2, (long long)(timeout_ms * 1e5)};
}

static void flag_wait(spp_ctx *ctx, hipStream_t st, const int *flag, int value, double timeout_ms)
{
	hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, st, make_flag_wait(ctx, flag, value, timeout_ms));
}

// Do the chain stream and the bulk stream run concurrently? Both directions are tried with the WAIT ENQUEUED
// FIRST: two streams that share a hardware queue would make a waiter block its own signaller; such a pair times
// out here (20 ms) and the event schedule stays in use for this ctx stream.
static bool flag_schedule_selftest(spp_ctx *ctx)
{
	DenseWork &dw = ctx->dense;
	hipStream_t st[2] = {ctx->stream, dw.aux};
	dw.sync.reserve(16);
	SPP_HIP_CHECK(hipMemsetAsync(dw.sync.p, 0, 16 * sizeof(int), ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	for(int p = 0; p < 2; ++ p) {
		flag_wait(ctx, st[p], dw.sync.p + p, 1, 20.0);
		flag_signal(st[1 - p], dw.sync.p + p, 1);
	}
	for(int i = 0; i < 2; ++ i)
		SPP_HIP_CHECK(hipStreamSynchronize(st[i]));
	int h_abort = 0;
	SPP_HIP_CHECK(hipMemcpy(&h_abort, dw.info.p + 2, sizeof(int), hipMemcpyDeviceToHost));
	if(h_abort) {
		SPP_HIP_CHECK(hipMemset(dw.info.p + 2, 0, sizeof(int)));
		return false;
	}
	return true;
}

static bool flag_schedule_usable(spp_ctx *ctx)
{
	static int sched = -1;
	if(sched < 0) {
		const char *e = getenv("SPP_DENSE_SCHED"); // 0: cross-stream events (round 1), 1: device flags
		sched = e ? atoi(e) : 1;
	}
	if(!sched)
		return false;
	DenseWork &dw = ctx->dense;
	if(dw.sync_state == 0 || dw.sync_stream != ctx->stream) {
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if(hipStreamIsCapturing(ctx->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
			return false; // no self-test (host synchronization) inside a capture
		dw.sync_state = flag_schedule_selftest(ctx) ? 1 : -1;
		dw.sync_stream = ctx->stream;
	}
	return dw.sync_state == 1;
}

static int dense_tail_rows()
{
	static int v = -1;
	if(v < 0) {
		const char *e = getenv("SPP_TAIL_ROWS"), *on = getenv("SPP_DENSE_TAIL");
		v = (on && !atoi(on)) ? 0 : (e ? atoi(e) : 44);
	}
	return v;
}

// the streamed tail only takes over a full factorization (every tile row a pivot block, no identity-padded pivots)
static bool dense_tail_applies(spp_ctx *ctx, int64_t rows, int64_t nsteps)
{
	return nsteps * NB >= rows && !(ctx->dense.ident_from >= 0 && ctx->dense.ident_from < rows);
}

// The streamed tail (spp_dense_tail.h): everything behind row panel k -- the update of step k, then steps k + 1 .. --
// as one launch, when the trailing matrix is a full factorization (every tile row a pivot block) of few enough tiles
// for one workgroup per CU. Returns false when it does not apply (the per-step schedule goes on).
static bool launch_dense_tail(spp_ctx *ctx, double *d_A, int64_t ld, int64_t rows, int64_t ncols, int64_t nsteps, int64_t k, bool has_rhs)
{
	static int enabled = -1;
	if(enabled < 0) {
		const char *e = getenv("SPP_DENSE_TAIL"); // 1: the streamed tail; 0: the per-step single-stream tail of round 2
		enabled = e ? atoi(e) : 1;
	}
	DenseWork &dw = ctx->dense;
	const int64_t c1 = (k + 1) * NB; // (k = -1: the whole factorization, no row panel in front of it)
	if(!enabled || k + 1 >= nsteps || nsteps * NB < rows || (dw.ident_from >= 0 && dw.ident_from < rows))
		return false;
	const int Tr = (int)(nsteps - (k + 1)), Tc = (int)((ncols - c1 + NB - 1) / NB);
	if(Tr < 1 || Tc < Tr || Tc > Tr + 1)
		return false;
	// (more tiles than CUs are fine: workgroups are dispatched row by row and leave after their own step, a workgroup that
	// starts late finds the row tiles of the steps it missed in memory and catches up at the speed of its matrix cores)
	const int ntile = Tr * Tc - Tr * (Tr - 1) / 2;
	static int max_rows = -1;
	if(max_rows < 0) {
		const char *e = getenv("SPP_TAIL_ROWS"); // tile rows from which on the factorization is streamed
		max_rows = e ? atoi(e) : 44;
	}
	if(Tr > max_rows)
		return false;
	hipStream_t s = ctx->stream;
	static uint64_t attr_seen = 0;
	if(first_on_this_device(attr_seen))
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)dense_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
			TAIL_LDS_DOUBLES * (int)sizeof(double)));
	if(dw.tail_pub.cap < (size_t)Tr * Tc) {
		dw.tail_pub.reserve(std::max<size_t>((size_t)Tr * Tc, 24 * 25));
		SPP_HIP_CHECK(hipMemsetAsync(dw.tail_pub.p, 0, dw.tail_pub.cap * sizeof(int), s));
		dw.tail_epoch = 0;
	}
	if(dw.tail_dinv.cap < (size_t)Tr * 8 * 256)
		dw.tail_dinv.reserve(std::max<size_t>((size_t)Tr * 8 * 256, (size_t)24 * 8 * 256));
	if(++ dw.tail_epoch >= (1 << 26)) { // (28 bits of epoch in a counter word)
		SPP_HIP_CHECK(hipMemsetAsync(dw.tail_pub.p, 0, dw.tail_pub.cap * sizeof(int), s));
		dw.tail_epoch = 1;
	}
	// Workgroup -> tile. Tile (i, j) needs row tiles of (k, i) and (k, j), k < i, so any key alpha i + beta j with alpha > 0,
	// beta >= 0 sorts the tiles topologically (producers before consumers: progress whatever is resident). Row by row
	// (beta = 0) the 256 CUs start on the first ~6 rows, far tiles included, and the near-diagonal tiles of the rows
	// behind them start late; beta > 0 holds the far columns back a little in favour of those.
	if(dw.tail_order_tr != Tr || dw.tail_order_tc != Tc) {
		static double beta = -1;
		if(beta < 0) {
			const char *e = getenv("SPP_TAIL_ORDER_BETA");
			beta = e ? atof(e) : 0.0;
		}
		std::vector<std::pair<double, int> > key;
		for(int i = 0; i < Tr; ++ i)
			for(int j = i; j < Tc; ++ j)
				key.push_back(std::make_pair((double)i + beta * (double)j, (i << 16) | j));
		std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x.first < y.first; });
		std::vector<int> order(key.size());
		for(size_t q = 0; q < key.size(); ++ q)
			order[q] = key[q].second;
		dw.tail_order.upload(order, s);
		dw.tail_order_tr = Tr;
		dw.tail_order_tc = Tc;
	}
	TailArgs a;
	a.order = dw.tail_order.p;
	a.A = d_A;
	a.ld = ld;
	a.rows = rows;
	a.ncols = ncols;
	a.c0 = c1;
	a.have_pre = (k >= 0) ? 1 : 0;
	a.Tr = Tr;
	a.Tc = Tc;
	a.has_rhs = has_rhs ? 1 : 0;
	a.tinv = dw.tinv_all.p + (size_t)(k + 1) * NB * NB;
	a.dbuf = dw.tail_dinv.p;
	a.pub = dw.tail_pub.p;
	a.epoch = dw.tail_epoch;
	a.info = dw.info.p;
	a.abort = dw.info.p + 2;
	a.timeout_ticks = (long long)(500.0 * 1e5);
	static int trace_env = -1;
	if(trace_env < 0) {
		const char *e = getenv("SPP_TAIL_TRACE"); // n: the n-th launch prints per tile row when its diagonal tile had all updates, was factored, and when the first panel tile started / ended
		trace_env = e ? atoi(e) : 0;
	}
	static int launches = 0;
	DevBuf<long long> trace;
	a.trace = nullptr;
	if(trace_env && ++ launches == trace_env) {
		trace.reserve((size_t)Tr * 8);
		SPP_HIP_CHECK(hipMemsetAsync(trace.p, 0, (size_t)Tr * 8 * sizeof(long long), s));
		a.trace = trace.p;
	}
	// (profiling: this launch is the dominant kernel of the factorization; algorithmic flops = the region's
	// factorization m^3 / 3, the update by the row panel in front of it, the right-hand side column)
	dom_begin(ctx);
	hipLaunchKernelGGL(dense_tail_kernel, dim3((unsigned)ntile), dim3(POTRF_THREADS), TAIL_LDS_DOUBLES * sizeof(double), s, a);
	{
		const double m = (double)(rows - c1);
		dom_end(ctx, m * m * m / 3.0 + (a.have_pre ? (double)NB * m * (m + 1.0) : 0.0) + 2.0 * m * m);
	}
	dw.tail_rows_last = Tr;
	SPP_HIP_CHECK(hipGetLastError());
	if(a.trace) {
		std::vector<long long> h((size_t)Tr * 8);
		SPP_HIP_CHECK(hipStreamSynchronize(s));
		SPP_HIP_CHECK(hipMemcpy(h.data(), trace.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
		fprintf(stderr, "[spp] streamed tail, %d tile rows (us from the launch's first stamp): diagonal tile: updates complete / factored+inverted | first panel tile: first row tile seen / last seen / last published\n", Tr);
		const long long t0 = h[0];
		for(int i = 0; i < Tr; ++ i)
			fprintf(stderr, "  row %2d  start %7.1f  updated %7.1f  done %7.1f | %7.1f %7.1f %7.1f\n", i, (h[8 * i] - t0) * 0.01, (h[8 * i + 1] - t0) * 0.01,
				(h[8 * i + 2] - t0) * 0.01, h[8 * i + 3] ? (h[8 * i + 3] - t0) * 0.01 : 0.0, h[8 * i + 4] ? (h[8 * i + 4] - t0) * 0.01 : 0.0, h[8 * i + 5] ? (h[8 * i + 5] - t0) * 0.01 : 0.0);
	}
	return true;
}

// The two-stream schedule. Chain (ctx stream): fused [tile row k+1 <- panel k, potrf_diag(k+1)], panel solve k+1.
// Bulk (CU-masked second stream): everything below the tile row, handed over right after the panel solve; once the
// trailing matrix is at most 2560 rows the whole step runs on the chain stream (one fused launch updates every
// trailing tile and factors the next diagonal block as soon as ITS tile is done).
// Measured in rounds 1-2 and no longer selectable (DESIGN.md section 3 records the numbers): handing the bulk update
// over after the tile row, waiting for it under the running potrf_diag, a slab-pipelined tile-row / panel kernel,
// pairs of steps as rank-256 updates, the rest of the tile row on a third stream beside potrf_diag, 64 x 64 and 64 x 32
// tiles for the single-stream update, skipping the zero half of the block inverse in the panel solve.
static void dense_factor_steps_enqueue(spp_ctx *ctx, double *d_A, int64_t ld, int64_t n, int64_t rows, int64_t ncols,
	int64_t nsteps, bool has_rhs, bool use_flags, bool allow_fused)
{
	hipStream_t s = ctx->stream, s2 = ctx->dense.aux;
	hipEvent_t evA = ctx->dense.ev[0], evB = ctx->dense.ev[1];
	// Hand-offs between the chain stream and the bulk stream: events (two barrier packets per direction, ~6 us
	// each on the critical chain) or, with use_flags, words in device memory: flag[2 k] = "row panel k is
	// complete" (signal kernel behind the panel solve on s, wait kernel in front of the bulk update on s2),
	// flag[2 k + 1] = "bulk update k is complete" (signal kernel behind it on s2; a wait kernel in front of the chain's
	// next tile-row kernel). Values are the epoch of this factorization.
	DenseWork &dw = ctx->dense;
	// pivots [n_id, n) are exact identity padding (big fronts of the sparse path pad their pivot block to a multiple of
	// 128): the diagonal-block kernel skips the 16-wide panels that consist of padding only
	const int64_t n_id = (dw.ident_from >= 0 && dw.ident_from < n) ? dw.ident_from : n;
	int ep = 0;
	int64_t step_a = -1, step_b = -1; // steps whose hand-over / bulk update the "events" currently name
	int64_t need_wait_b = -1;         // flags only: the next chain kernel has to wait for bulk update need_wait_b
	if(use_flags) {
		if(dw.sync.cap < (size_t)(2 * nsteps + 16)) {
			dw.sync.reserve((size_t)(2 * nsteps + 16));
			SPP_HIP_CHECK(hipMemsetAsync(dw.sync.p, 0, dw.sync.cap * sizeof(int), s));
			dw.sync_epoch = 1; // the self-test used the value 1
		}
		ep = ++ dw.sync_epoch;
		// the bulk stream must start after everything enqueued on the ctx stream so far (the matrix itself): one event per factorization
		SPP_HIP_CHECK(hipEventRecord(evA, s));
		SPP_HIP_CHECK(hipStreamWaitEvent(s2, evA, 0));
	}
	int64_t pend_sig = -1; // flags only: "row panel pend_sig is complete" not yet enqueued (merged with the next wait on s)
	auto record_a = [&](int64_t k) { // row panel k is complete on s
		step_a = k;
		if(use_flags)
			pend_sig = k; // enqueued by flush_wait_b(), together with the wait that follows it on the chain stream
		else
			SPP_HIP_CHECK(hipEventRecord(evA, s));
	};
	// (flags) the bulk stream's "bulk update k is complete" is not enqueued on its own either: the wait for the next row
	// panel follows it directly on that stream, and the two are one launch -- a one-wave kernel costs 5-6 us of queue time
	// each, between two bulk updates that otherwise run back to back
	int64_t pend_sig_b = -1;
	auto wait_a_on_bulk = [&]() {
		if(use_flags && pend_sig_b >= 0) {
			hipLaunchKernelGGL(flag_signal_wait_kernel, dim3(1), dim3(64), 0, s2, dw.sync.p + 2 * pend_sig_b + 1, ep,
				make_flag_wait(ctx, dw.sync.p + 2 * step_a, ep));
			pend_sig_b = -1;
		} else if(use_flags)
			flag_wait(ctx, s2, dw.sync.p + 2 * step_a, ep);
		else
			SPP_HIP_CHECK(hipStreamWaitEvent(s2, evA, 0));
	};
	auto flush_sig_b = [&]() {
		if(pend_sig_b >= 0)
			flag_signal(s2, dw.sync.p + 2 * pend_sig_b + 1, ep);
		pend_sig_b = -1;
	};
	auto record_b = [&](int64_t k) { // bulk update k is complete on s2
		step_b = k;
		if(use_flags)
			pend_sig_b = k; // enqueued with the wait for row panel k + 1 (wait_a_on_bulk), or by flush_sig_b()
		else
			SPP_HIP_CHECK(hipEventRecord(evB, s2));
	};
	auto wait_b_now = [&]() { // a wait of its own on the chain stream (event or one-wave kernel)
		if(use_flags && pend_sig_b == step_b)
			flush_sig_b(); // (no further hand-over to the bulk stream took the signal along)
		if(use_flags && pend_sig >= 0) { // a deferred panel signal goes first (never left behind a wait)
			flag_signal(s, dw.sync.p + 2 * pend_sig, ep);
			pend_sig = -1;
		}
		if(use_flags)
			flag_wait(ctx, s, dw.sync.p + 2 * step_b + 1, ep);
		else
			SPP_HIP_CHECK(hipStreamWaitEvent(s, evB, 0));
	};
	auto wait_b_deferred = [&]() { // in front of the chain's next tile-row kernel (flags), or an event wait now
		if(use_flags)
			need_wait_b = step_b;
		else
			SPP_HIP_CHECK(hipStreamWaitEvent(s, evB, 0));
	};
	auto flush_wait_b = [&]() { // everything deferred on the chain stream: the panel signal, the wait for a bulk update, or both in one launch
		if(need_wait_b >= 0 && pend_sig_b == need_wait_b)
			flush_sig_b(); // the signal this wait is for must be in the bulk stream's queue
		if(need_wait_b >= 0 && pend_sig >= 0)
			hipLaunchKernelGGL(flag_signal_wait_kernel, dim3(1), dim3(64), 0, s, dw.sync.p + 2 * pend_sig, ep,
				make_flag_wait(ctx, dw.sync.p + 2 * need_wait_b + 1, ep));
		else if(need_wait_b >= 0)
			flag_wait(ctx, s, dw.sync.p + 2 * need_wait_b + 1, ep);
		else if(pend_sig >= 0)
			flag_signal(s, dw.sync.p + 2 * pend_sig, ep);
		need_wait_b = pend_sig = -1;
	};
	bool bulk_pending = false;
	// Fused chain kernel (update_potrf_kernel): the update of a row region from panel kp and the factorization of the
	// region's first diagonal block (step kn) in one launch; the panel solve of step kn follows as its own launch.
	static int fused_env = -1;
	if(fused_env < 0) {
		const char *e = getenv("SPP_FUSED"); // 0: tile row / update, potrf_diag and panel solve as three launches (round 1)
		fused_env = e ? atoi(e) : 1;
	}
	const int fused = allow_fused ? fused_env : 0; // (its sub-tile counters are monotonic: not inside a stream capture)
	if(fused) {
		static uint64_t fattr_seen = 0;
		if(first_on_this_device(fattr_seen)) {
			SPP_HIP_CHECK(hipFuncSetAttribute((const void*)update_potrf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
				POTRF_LDS_DOUBLES * (int)sizeof(double)));
		}
		// per-step counters of finished diagonal sub-tiles: monotonic, never reset between factorizations (the host keeps
		// the value each slot will have reached) -- a memset per call was a 4.5 us launch in front of every big front
		if(dw.fuse_cnt.cap < (size_t)(nsteps + 2) || dw.fuse_dirty) {
			dw.fuse_cnt.reserve(std::max<size_t>((size_t)(nsteps + 2), 64));
			SPP_HIP_CHECK(hipMemsetAsync(dw.fuse_cnt.p, 0, dw.fuse_cnt.cap * sizeof(int), s));
			dw.fuse_expect.assign(dw.fuse_cnt.cap, 0);
			dw.fuse_dirty = false;
		}
	}
	static_assert(POTRF_LDS_DOUBLES >= (64 + 64) * FS_STRIDE, "the fused kernel's LDS is sized by the factorization");
	// region rows [r0, r0 + m) x cols [r0, ncols), panel rows at kp0; potrf of step kn when kn >= 0
	auto fused_update_potrf = [&](int64_t r0, int64_t m, int64_t kp0, int64_t kn) {
		const int64_t N = ncols - r0;
		if(m <= 0 || N <= 0)
			return false;
		const int64_t md = std::min<int64_t>(NB, m), nd = std::min<int64_t>(NB, N);
		const int ti = (int)((md + 31) / 32), tj = (int)((nd + 31) / 32);
		int nA = 0;
		for(int j = 0; j < tj; ++ j)
			nA += std::min(j, ti - 1) + 1;
		const int64_t nbi = (m + 63) / 64, nbj = (N + 63) / 64;
		const int do_potrf = kn >= 0 ? 1 : 0;
		const int64_t kn0 = (kn >= 0 ? kn : 0) * NB;
		const int n_valid = (int)std::max<int64_t>(0, std::min<int64_t>(NB, n_id - kn0));
		const size_t slot = (size_t)(kn >= 0 ? kn : nsteps + 1);
		dw.fuse_expect[slot] += nA;
		hipLaunchKernelGGL(update_potrf_kernel, dim3((unsigned)(nA + do_potrf + nbi * nbj)), dim3(POTRF_THREADS),
			POTRF_LDS_DOUBLES * sizeof(double), s, m, N, d_A + kp0 + r0 * ld, ld, d_A + r0 + r0 * ld, nA, dw.fuse_expect[slot], do_potrf,
			n_valid, (has_rhs && n_valid < NB) ? 1 : 0, dw.tinv_all.p + (size_t)(kn >= 0 ? kn : 0) * NB * NB, dw.info.p, kn0,
			dw.fuse_cnt.p + slot, dw.info.p + 2, (long long)(500.0 * 1e5));
		return true;
	};
	auto potrf_and_panel = [&](int64_t k, bool potrf_done = false) {
		const int64_t k0 = k * NB;
		const int n_valid = (int)std::max<int64_t>(0, std::min<int64_t>(NB, n_id - k0));
		double *tinv = ctx->dense.tinv_all.p + (size_t)k * NB * NB;
		if(!potrf_done)
			hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_DOUBLES * sizeof(double), s,
				d_A + k0 + k0 * ld, ld, n_valid, (has_rhs && n_valid < NB) ? 1 : 0, tinv, ctx->dense.info.p, k0);
		const int64_t c1 = k0 + NB;
		if(c1 < ncols) // panel: R_kj = R_kk^-T S_kj in place, 16-column slabs, one MFMA tile per wave
			launch_panel_solve(s, ncols - c1, tinv, d_A + k0 + c1 * ld, ld);
	};
	auto tile_row = [&](int64_t r0, int64_t kp0) { // rows [r0, r0 + 128) x cols [r0, ncols) -= P^T P, panel at kp0
		const int64_t m = std::min<int64_t>(NB, rows - r0);
		if(m <= 0 || r0 >= ncols)
			return;
		const double *P = d_A + kp0 + r0 * ld;
		flush_wait_b();
		launch_gemm_staged<32, 32, 16, 16, 0>(s, m, ncols - r0, NB, P, ld, P, ld, d_A + r0 + r0 * ld, ld, true);
	};
	if(allow_fused && nsteps >= 2 && dense_tail_rows() * NB >= rows && dense_tail_applies(ctx, rows, nsteps) &&
		launch_dense_tail(ctx, d_A, ld, rows, ncols, nsteps, -1, has_rhs)) {
		dw.tinv_half = 0; // the whole factorization streamed: every block inverse is complete
		SPP_HIP_CHECK(hipGetLastError());
		return;
	}
	potrf_and_panel(0);
	const int64_t single_below = 2560, single_mixed_above = 1280;
	for(int64_t k = 0; k < nsteps; ++ k) {
		const int64_t k0 = k * NB, c1 = k0 + NB, c2 = c1 + NB;
		if(c1 >= ncols || rows - c1 <= 0)
			break;
		if(rows - c1 <= single_below || (allow_fused && dense_tail_rows() * NB >= rows - c1 && dense_tail_applies(ctx, rows, nsteps))) {
			// small trailing matrix: the whole update is shorter than a cross-stream hand-off plus the tile row
			if(bulk_pending) {
				wait_b_now();
				bulk_pending = false;
			}
			const double *P = d_A + k0 + c1 * ld;
			if(allow_fused && launch_dense_tail(ctx, d_A, ld, rows, ncols, nsteps, k, has_rhs)) {
				dw.tinv_half = k + 1; // (the streamed tail stores whole block inverses)
				break;
			}
			if(fused && k + 1 < nsteps && rows - c1 >= NB) {
				// one launch: every trailing tile <- panel k, the next diagonal block factored as soon as ITS tile is done
				fused_update_potrf(c1, rows - c1, k0, k + 1);
				potrf_and_panel(k + 1, true);
				continue;
			}
			if(rows - c1 >= single_mixed_above) { // the fine-grained MFMA kernel wins from ~10 tile rows on
				dom_begin(ctx);
				const bool big = dense_gemm_tn_sub(ctx, rows - c1, ncols - c1, NB, P, ld, P, ld, d_A + c1 + c1 * ld, ld, true);
				const double mr = (double)(rows - c1);
				if(big)
					dom_end(ctx, 2.0 * NB * (0.5 * mr * (mr + 1.0) + mr * (double)(ncols - rows)));
			} else
				launch_gemm_staged<32, 32, 16, 16, 0>(s, rows - c1, ncols - c1, NB, P, ld, P, ld, d_A + c1 + c1 * ld, ld, true);
			if(k + 1 < nsteps)
				potrf_and_panel(k + 1);
			continue;
		}
		// The bulk update needs only row panel k (complete at this point of the chain stream) and writes rows >= c2,
		// disjoint from the tile row the chain touches next: it is handed to the bulk stream BEFORE the tile row, so
		// that consecutive bulk updates run back to back.
		record_a(k);
		if(bulk_pending) { // the previous bulk update touched every row >= c1
			wait_b_deferred();
			bulk_pending = false;
		}
		const bool have_bulk = rows - c2 > 0 && c2 < ncols;
		bool fused_done = false; // potrf of step k+1 already done by the fused kernel
		if(have_bulk) {
			wait_a_on_bulk();
			const double *P = d_A + k0 + c2 * ld;
			hipStream_t keep = ctx->stream;
			ctx->stream = s2; // dom events + gemm launch on the bulk stream
			dom_begin(ctx);
			const bool big = dense_gemm_tn_sub(ctx, rows - c2, ncols - c2, NB, P, ld, P, ld, d_A + c2 + c2 * ld, ld, true);
			const double mr = (double)(rows - c2);
			// useful flops: upper triangle of the M x M part + the (N - M) extra columns (rhs); only the
			// launches of the 128 x 128-tile kernel are accounted (one kernel symbol = one rocprof row)
			if(big)
				dom_end(ctx, 2.0 * NB * (0.5 * mr * (mr + 1.0) + mr * (double)(ncols - rows)));
			ctx->stream = keep;
			record_b(k);
			bulk_pending = true;
		}
		if(fused && k + 1 < nsteps && rows - c1 >= NB) {
			flush_wait_b();
			fused_update_potrf(c1, NB, k0, k + 1); // tile row k+1 <- panel k, potrf(k+1) inside
			fused_done = true;
		} else
			tile_row(c1, k0);
		flush_wait_b();
		if(k + 1 < nsteps)
			potrf_and_panel(k + 1, fused_done);
	}
	if(bulk_pending) { // the join is an event: the ctx stream's successors need the whole bulk stream drained
		pend_sig_b = -1; // (nobody waits for that flag)
		if(use_flags)
			SPP_HIP_CHECK(hipEventRecord(evB, s2));
		SPP_HIP_CHECK(hipStreamWaitEvent(s, evB, 0));
	}
	SPP_HIP_CHECK(hipGetLastError());
}

void dense_reserve(spp_ctx *ctx, int64_t nblk)
{
	ensure_dense_work(ctx, nblk);
}

void dense_info_reset(spp_ctx *ctx)
{
	ensure_info(ctx);
	hipLaunchKernelGGL(set_info_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dense.info.p);
}

void dense_chain_check(spp_ctx *ctx)
{
	if(ctx->dense.h_chain_err && *ctx->dense.h_chain_err) {
		*ctx->dense.h_chain_err = 0;
		throw Error(SPP_E_HIP, "backward substitution: a workgroup of the chain kernel timed out waiting for its predecessor");
	}
}

int dense_info_fetch(spp_ctx *ctx, bool *dag_aborted)
{
	int h_info[4] = {0, 0, 0, 0};
	SPP_HIP_CHECK(hipMemcpyAsync(h_info, ctx->dense.info.p, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if(h_info[3]) { // a front of the dependency-driven sparse launches timed out waiting for a child / its parent
		SPP_HIP_CHECK(hipMemset(ctx->dense.info.p + 3, 0, sizeof(int)));
		sparse_dag_disable(ctx);
		if(dag_aborted) { // the caller repeats the solve level by level
			*dag_aborted = true;
			return 0;
		}
		throw Error(SPP_E_HIP, "sparse factorization: a front timed out waiting for another front's flag; "
			"the following calls launch level by level");
	}
	if(h_info[2] && h_info[0]) {
		// the streamed tail raises the abort word when a diagonal block is not positive definite (its consumers give up
		// waiting): that is a failed factorization, not a lost hand-over
		SPP_HIP_CHECK(hipMemset(ctx->dense.info.p + 2, 0, sizeof(int)));
		return h_info[0];
	}
	if(h_info[2]) { // a cross-stream flag wait timed out: the result is garbage, the flag hand-offs stay off
		SPP_HIP_CHECK(hipMemset(ctx->dense.info.p + 2, 0, sizeof(int)));
		ctx->dense.sync_state = -1;
		ctx->dense.fuse_dirty = true; // the sub-tile counters of the aborted factorization are behind the host's bookkeeping
		throw Error(SPP_E_HIP, "dense factorization: a cross-stream flag wait timed out (streams not concurrent?); "
			"the following calls use the event schedule");
	}
	return h_info[0];
}

// factorization without the host round trip for the status: the caller enqueues what follows (solves, back-
// substitution) and fetches the status with dense_info_fetch() at the end -- after a failed factorization
// those kernels work on garbage, which is harmless (no data-dependent waits) and discarded by the caller
void dense_potrf_upper_enqueue(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld)
{
	SPP_REQUIRE(ld % NB == 0 && n < ld, SPP_E_BADARG, "dense_potrf_upper: ld must be a multiple of 128 and > n");
	dense_info_reset(ctx);
	dense_factor_steps(ctx, d_A, ld, n, n, n + 1, (n + NB - 1) / NB, true);
}

int dense_potrf_upper(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld, bool /*keep_inverses*/)
{
	SPP_REQUIRE(ld % NB == 0 && n < ld, SPP_E_BADARG, "dense_potrf_upper: ld must be a multiple of 128 and > n");
	dense_info_reset(ctx);
	dense_factor_steps(ctx, d_A, ld, n, n, n + 1, (n + NB - 1) / NB, true);
	return dense_info_fetch(ctx) ? SPP_NOT_POSDEF : SPP_OK;
}

// back substitution R x = y with y in d_b (n entries); uses the block inverses of the last potrf.
// d_b may be the rhs column of the factored matrix itself.
void dense_potrs_upper(spp_ctx *ctx, const double *d_R, int64_t n, int64_t ld, double *d_b)
{
	const int64_t nblk = (n + NB - 1) / NB;
	hipStream_t s = ctx->stream;
	if(ctx->dense.tinv_half > 0) { // the factorization left [T0 R01; 0 T1] per block: whole inverses from here on
		hipLaunchKernelGGL(tinv_finish_kernel, dim3((unsigned)ctx->dense.tinv_half), dim3(1024), 0, s, ctx->dense.tinv_all.p);
		ctx->dense.tinv_half = 0;
	}
	static int chain = -1;
	if(chain < 0) {
		const char *e = getenv("SPP_TRSV_CHAIN"); // 2: the chain inside one workgroup (default), 1: a workgroup per hop (round 2), 0: a launch per hop (round 1)
		chain = e ? atoi(e) : 2;
	}
	int use_chain = chain;
	{
		// both chain kernels tell one solve from the next by a kernel argument (epoch / check constant): a captured
		// launch would replay it stale -- inside a stream capture the substitution is a launch per block row
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if(hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
			use_chain = 0;
	}
	if(use_chain == 2 && nblk > 1 && nblk <= 4096) {
		DenseWork &dw = ctx->dense;
		const size_t need = (size_t)nblk * NB * 4; // x and w, two words per element
		if(dw.trsv_pay.cap < need) {
			dw.trsv_pay.reserve(need);
			SPP_HIP_CHECK(hipMemsetAsync(dw.trsv_pay.p, 0, dw.trsv_pay.cap * sizeof(double), s));
		}
		if(!dw.h_chain_err) {
			SPP_HIP_CHECK(hipHostMalloc((void**)&dw.h_chain_err, sizeof(int), hipHostMallocDefault));
			*dw.h_chain_err = 0;
		}
		++ dw.epoch;
		// the check constant of this solve: odd multiples of a 64-bit odd constant are distinct and non-zero for 2^63 solves
		const unsigned long long K = (2ull * (unsigned long long)(unsigned)dw.epoch + 1ull) * 0x9E3779B97F4A7C15ull;
		TrsvPay *xpay = (TrsvPay*)dw.trsv_pay.p, *wpay = xpay + (size_t)nblk * NB;
		static int mform = -1;
		if(mform < 0) {
			const char *e = getenv("SPP_TRSV_MFORM"); // 0: the chain applies R_{b, b+1} and Tinv_b itself (two tiles per hop)
			mform = e ? atoi(e) : 1;
		}
		if(mform) {
			if(dw.trsv_m.cap < (size_t)nblk * NB * NB)
				dw.trsv_m.reserve((size_t)nblk * NB * NB);
			static uint64_t attr_seen = 0;
			if(first_on_this_device(attr_seen))
				SPP_HIP_CHECK(hipFuncSetAttribute((const void*)trsv_m_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TRSVM_LDS_DOUBLES * (int)sizeof(double)));
			hipLaunchKernelGGL(trsv_m_kernel, dim3((unsigned)nblk - 1, 8), dim3(1024), TRSVM_LDS_DOUBLES * sizeof(double), s, d_R, ld, n, dw.tinv_all.p, dw.trsv_m.p);
			hipLaunchKernelGGL((trsv_back_chain2_kernel<8, true>), dim3((unsigned)nblk + 1), dim3(512), 0, s,
				d_R, ld, n, (int)nblk, dw.tinv_all.p, dw.trsv_m.p, d_b, xpay, wpay, K, dw.info.p + 1);
		} else
			hipLaunchKernelGGL((trsv_back_chain2_kernel<8, false>), dim3((unsigned)nblk + 1), dim3(512), 0, s,
				d_R, ld, n, (int)nblk, dw.tinv_all.p, (const double*)nullptr, d_b, xpay, wpay, K, dw.info.p + 1);
		SPP_HIP_CHECK(hipGetLastError());
		SPP_HIP_CHECK(hipMemcpyAsync(dw.h_chain_err, dw.info.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
		return;
	}
	if(use_chain && nblk > 1 && nblk <= 1024) {
		// one launch, the dependent chain runs through device flags (info[1] = timeout flag)
		DenseWork &dw = ctx->dense;
		if(dw.flags.cap < (size_t)nblk) {
			dw.flags.reserve((size_t)nblk);
			SPP_HIP_CHECK(hipMemsetAsync(dw.flags.p, 0, (size_t)nblk * sizeof(int), s));
			dw.epoch = 0;
		}
		if(!dw.h_chain_err) {
			SPP_HIP_CHECK(hipHostMalloc((void**)&dw.h_chain_err, sizeof(int), hipHostMallocDefault));
			*dw.h_chain_err = 0;
		}
		++ dw.epoch;
		hipLaunchKernelGGL(trsv_back_chain_kernel, dim3((unsigned)nblk), dim3(256), 0, s, d_R, ld, n, (int)nblk,
			dw.tinv_all.p, d_b, dw.xtmp.p, dw.flags.p, dw.epoch, dw.info.p + 1);
		SPP_HIP_CHECK(hipGetLastError());
		SPP_HIP_CHECK(hipMemcpyAsync(d_b, dw.xtmp.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
		// the timeout flag travels to pinned host memory; dense_chain_check() looks at it after the
		// next synchronization of the stream (no extra round trip)
		SPP_HIP_CHECK(hipMemcpyAsync(dw.h_chain_err, dw.info.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
		return;
	}
	for(int64_t k = nblk; k > 0;) {
		-- k;
		const int64_t k0 = k * NB;
		const double *tinv = ctx->dense.tinv_all.p + (size_t)k * NB * NB;
		unsigned nwg = 1 + (unsigned)((k0 + 255) / 256);
		const int nv = (int)((n - k0 < NB) ? (n - k0) : NB);
		hipLaunchKernelGGL(trsv_back_kernel, dim3(nwg), dim3(256), 0, s, d_R, ld, k0, nv, tinv, d_b, ctx->dense.xtmp.p);
	}
	SPP_HIP_CHECK(hipGetLastError());
	SPP_HIP_CHECK(hipMemcpyAsync(d_b, ctx->dense.xtmp.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
}

// --------------------------------------------------------------------------------------------------
// micro-benchmarks: measured peaks reported beside the spec peaks in bench.py
// --------------------------------------------------------------------------------------------------
__global__ void copy_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	for(; i < n; i += stride)
		dst[i] = src[i];
}

double microbench_copy(spp_ctx *ctx, size_t bytes, int iters)
{
	DevBuf<double2> a, b;
	size_t n = bytes / sizeof(double2);
	a.reserve(n);
	b.reserve(n);
	SPP_HIP_CHECK(hipMemsetAsync(a.p, 0, n * sizeof(double2), ctx->stream));
	hipEvent_t e0, e1;
	SPP_HIP_CHECK(hipEventCreate(&e0));
	SPP_HIP_CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, ctx->stream, a.p, b.p, n);
	SPP_HIP_CHECK(hipEventRecord(e0, ctx->stream));
	for(int i = 0; i < iters; ++ i)
		hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, ctx->stream, a.p, b.p, n);
	SPP_HIP_CHECK(hipEventRecord(e1, ctx->stream));
	SPP_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	SPP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return 2.0 * n * sizeof(double2) * iters / (ms * 1e-3) * 1e-9;
}

// The trailing-update kernel's C-tile access pattern in isolation (8 B per lane, four 128-byte column
// segments per wave instruction, 128x128 tile per 1024-thread workgroup): reads C and writes -C back.
// A known byte count in exactly this pattern calibrates FETCH_SIZE / WRITE_SIZE for that kernel.
__global__ __launch_bounds__(1024)
void ctile_rw_kernel(double *C, int64_t ldc)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int l15 = lane & 15, l4 = lane >> 4;
	const int64_t m0 = (int64_t)blockIdx.x * 128 + (wave % 4) * 32, n0 = (int64_t)blockIdx.y * 128 + (wave / 4) * 32;
	double v[2][2][4];
#pragma unroll
	for(int b = 0; b < 2; ++ b)
#pragma unroll
		for(int a = 0; a < 2; ++ a)
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				v[b][a][r] = C[m0 + a * 16 + l15 + (n0 + b * 16 + l4 + 4 * r) * ldc];
#pragma unroll
	for(int b = 0; b < 2; ++ b)
#pragma unroll
		for(int a = 0; a < 2; ++ a)
#pragma unroll
			for(int r = 0; r < 4; ++ r)
				C[m0 + a * 16 + l15 + (n0 + b * 16 + l4 + 4 * r) * ldc] = -v[b][a][r];
}

double microbench_ctile(spp_ctx *ctx, int n, int iters)
{
	SPP_REQUIRE(n > 0 && n % 128 == 0, SPP_E_BADARG, "ctile microbenchmark: n must be a multiple of 128");
	DevBuf<double> c;
	c.reserve((size_t)n * n);
	SPP_HIP_CHECK(hipMemsetAsync(c.p, 0, (size_t)n * n * sizeof(double), ctx->stream));
	hipEvent_t e0, e1;
	SPP_HIP_CHECK(hipEventCreate(&e0));
	SPP_HIP_CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(ctile_rw_kernel, dim3(n / 128, n / 128), dim3(1024), 0, ctx->stream, c.p, (int64_t)n);
	SPP_HIP_CHECK(hipEventRecord(e0, ctx->stream));
	for(int i = 0; i < iters; ++ i)
		hipLaunchKernelGGL(ctile_rw_kernel, dim3(n / 128, n / 128), dim3(1024), 0, ctx->stream, c.p, (int64_t)n);
	SPP_HIP_CHECK(hipEventRecord(e1, ctx->stream));
	SPP_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	SPP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return 2.0 * n * n * sizeof(double) * iters / (ms * 1e-3) * 1e-9;
}

// Issue rate of v_mfma_f64_16x16x4 with the operand pattern of a real tile product: a 2 x 2 register outer product fed
// from eight + two distinct operand registers holding data (tools/mfma_rate2.hip). One operand pair reused by every
// instruction -- the loop of rounds 1-2 -- measures 46-48 TFLOP/s on the same part; this pattern 67.
__global__ __launch_bounds__(256)
void mfma_f64_peak_kernel(double *out, int iters)
{
	double fn[8], fm[2];
#pragma unroll
	for(int i = 0; i < 8; ++ i)
		fn[i] = 0.25 + 1e-3 * ((threadIdx.x * 7 + i * 13) & 63);
#pragma unroll
	for(int i = 0; i < 2; ++ i)
		fm[i] = 0.5 - 1e-3 * ((threadIdx.x * 5 + i * 11 + blockIdx.x) & 63);
	v4f64 acc[2][2];
#pragma unroll
	for(int i = 0; i < 2; ++ i)
#pragma unroll
		for(int j = 0; j < 2; ++ j)
			acc[i][j] = (v4f64){0, 0, 0, 0};
	for(int it = 0; it < iters; ++ it) {
#pragma unroll
		for(int r = 0; r < 4; ++ r)
#pragma unroll
			for(int i = 0; i < 2; ++ i)
#pragma unroll
				for(int j = 0; j < 2; ++ j)
					acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[i + 2 * r], fm[j], acc[i][j], 0, 0, 0);
		fm[0] = -fm[0]; // (the sums stay bounded, the operands keep changing)
	}
	double s = 0;
#pragma unroll
	for(int i = 0; i < 2; ++ i)
#pragma unroll
		for(int j = 0; j < 2; ++ j)
			s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

double microbench_mfma_f64(spp_ctx *ctx, int iters)
{
	const int nblk = 256 * 8;
	DevBuf<double> out;
	out.reserve((size_t)nblk * 256);
	hipEvent_t e0, e1;
	SPP_HIP_CHECK(hipEventCreate(&e0));
	SPP_HIP_CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out.p, 16);
	SPP_HIP_CHECK(hipEventRecord(e0, ctx->stream));
	hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out.p, iters);
	SPP_HIP_CHECK(hipEventRecord(e1, ctx->stream));
	SPP_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	SPP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	// per wave: iters * 16 MFMAs of 16*16*4*2 flops
	double flops = (double)nblk * 4 * (double)iters * 16 * 2048.0;
	return flops / (ms * 1e-3) * 1e-12;
}

// the bulk trailing update of one factorization step, stand-alone: C (m x (m + 1), upper tiles) -= P^T P with a
// 128-row panel P, `iters` launches back to back, hipEvents on the ctx stream; returns ms per launch
double microbench_update(spp_ctx *ctx, int64_t m, int iters)
{
	const int64_t ld = ((m + 1 + 127) / 128) * 128;
	DevBuf<double> p, c;
	p.reserve((size_t)ld * 128);
	c.reserve((size_t)ld * ld);
	SPP_HIP_CHECK(hipMemsetAsync(p.p, 0, (size_t)ld * 128 * sizeof(double), ctx->stream));
	SPP_HIP_CHECK(hipMemsetAsync(c.p, 0, (size_t)ld * ld * sizeof(double), ctx->stream));
	hipEvent_t e0, e1;
	SPP_HIP_CHECK(hipEventCreate(&e0));
	SPP_HIP_CHECK(hipEventCreate(&e1));
	// the panel is stored as the rows [0, 128) of a 128 x (m + 1) strip with leading dimension 128
	dense_gemm_tn_sub(ctx, m, m + 1, 128, p.p, 128, p.p, 128, c.p, ld, true);
	SPP_HIP_CHECK(hipEventRecord(e0, ctx->stream));
	for(int i = 0; i < iters; ++ i)
		dense_gemm_tn_sub(ctx, m, m + 1, 128, p.p, 128, p.p, 128, c.p, ld, true);
	SPP_HIP_CHECK(hipEventRecord(e1, ctx->stream));
	SPP_HIP_CHECK(hipEventSynchronize(e1));
	float ms = 0;
	SPP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return ms / iters;
}

} // namespace spp
