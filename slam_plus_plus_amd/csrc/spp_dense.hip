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

namespace spp {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;           // k-slab staged through LDS
constexpr int LDS_STRIDE = 18;   // doubles per tile column in LDS: conflict-free ds_read_b64, 16-B aligned

// --------------------------------------------------------------------------------------------------
// C (M x N) {-=, =} A^T B,  A: K x M (lda), B: K x N (ldb), K % 16 == 0. Column-major.
// MODE 0: C -= A^T B (trailing update);  MODE 1: C = A^T B (may alias B when BM covers all rows).
// --------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64)
void gemm_tn_kernel(int64_t M, int64_t N, int K, const double *__restrict__ A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, int upper_only)
{
	constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
	constexpr int TA = WM / 16, TB = WN / 16;       // MFMA tiles per wave
	constexpr int PA = (BM * 8) / NT, PB = (BN * 8) / NT; // 16-byte pieces per thread per slab
	static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");

	const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
	if(upper_only && m0 >= n0 + BN)
		return; // tile strictly below the diagonal

	__shared__ double As[BM * LDS_STRIDE];
	__shared__ double Bs[BN * LDS_STRIDE];

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = (wave % NWM) * WM, wn = (wave / NWM) * WN;
	const int l15 = lane & 15, l4 = lane >> 4;

	v4f64 acc[TB][TA];
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a)
			acc[b][a] = (v4f64){0, 0, 0, 0};

	// global staging: piece p -> (column p / 8, 16-byte piece p % 8)
	double2 ra[PA], rb[PB];
	const double *pa[PA], *pb[PB];
#pragma unroll
	for(int i = 0; i < PA; ++ i) {
		int p = tid + i * NT;
		int64_t col = m0 + (p >> 3);
		if(col > M - 1) col = M - 1; // clamp: computes garbage that is never stored
		pa[i] = A + col * lda + (p & 7) * 2;
	}
#pragma unroll
	for(int i = 0; i < PB; ++ i) {
		int p = tid + i * NT;
		int64_t col = n0 + (p >> 3);
		if(col > N - 1) col = N - 1;
		pb[i] = B + col * ldb + (p & 7) * 2;
	}
#pragma unroll
	for(int i = 0; i < PA; ++ i)
		ra[i] = *(const double2*)(pa[i]);
#pragma unroll
	for(int i = 0; i < PB; ++ i)
		rb[i] = *(const double2*)(pb[i]);

	for(int k0 = 0; k0 < K; k0 += BK) {
		__syncthreads(); // previous slab fully consumed
#pragma unroll
		for(int i = 0; i < PA; ++ i) {
			int p = tid + i * NT;
			*(double2*)(&As[(p >> 3) * LDS_STRIDE + (p & 7) * 2]) = ra[i];
		}
#pragma unroll
		for(int i = 0; i < PB; ++ i) {
			int p = tid + i * NT;
			*(double2*)(&Bs[(p >> 3) * LDS_STRIDE + (p & 7) * 2]) = rb[i];
		}
		__syncthreads();
		if(k0 + BK < K) { // prefetch the next slab into registers while the MFMAs run
#pragma unroll
			for(int i = 0; i < PA; ++ i)
				ra[i] = *(const double2*)(pa[i] + k0 + BK);
#pragma unroll
			for(int i = 0; i < PB; ++ i)
				rb[i] = *(const double2*)(pb[i] + k0 + BK);
		}
#pragma unroll
		for(int kk = 0; kk < BK / 4; ++ kk) {
			double fa[TA], fb[TB];
#pragma unroll
			for(int a = 0; a < TA; ++ a)
				fa[a] = As[(wm + a * 16 + l15) * LDS_STRIDE + kk * 4 + l4];
#pragma unroll
			for(int b = 0; b < TB; ++ b)
				fb[b] = Bs[(wn + b * 16 + l15) * LDS_STRIDE + kk * 4 + l4];
#pragma unroll
			for(int b = 0; b < TB; ++ b)
#pragma unroll
				for(int a = 0; a < TA; ++ a)
					acc[b][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[b][a], 0, 0, 0);
		}
	}
	if(MODE == 1)
		__syncthreads(); // in-place: every wave has finished reading B before anyone writes C

	// D[row = l4 + 4 r][col = l15] of tile (b, a) is C(m = wm + 16 a + l15, n = wn + 16 b + l4 + 4 r)
#pragma unroll
	for(int b = 0; b < TB; ++ b)
#pragma unroll
		for(int a = 0; a < TA; ++ a) {
			const int64_t m = m0 + wm + a * 16 + l15;
#pragma unroll
			for(int r = 0; r < 4; ++ r) {
				const int64_t n = n0 + wn + b * 16 + l4 + 4 * r;
				if(m < M && n < N) {
					double *c = C + m + n * ldc;
					if(MODE == 0)
						*c -= acc[b][a][r];
					else
						*c = acc[b][a][r];
				}
			}
		}
}

template <int BM, int BN, int WM, int WN, int MODE>
static void launch_gemm(hipStream_t s, int64_t M, int64_t N, int K, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only)
{
	dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
	dim3 block((BM / WM) * (BN / WN) * 64);
	if(!grid.x || !grid.y)
		return;
	hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN, MODE>), grid, block, 0, s,
		M, N, K, A, lda, B, ldb, C, ldc, upper_only ? 1 : 0);
}

void dense_gemm_tn_sub(spp_ctx *ctx, int64_t m, int64_t n, int64_t k, const double *A, int64_t lda,
	const double *B, int64_t ldb, double *C, int64_t ldc, bool upper_only)
{
	SPP_REQUIRE(k % BK == 0, SPP_E_BADARG, "gemm_tn_sub: k must be a multiple of 16");
	if(!m || !n || !k)
		return;
	// 128 x 128 tiles when they fill the chip, 64 x 64 tiles for the tail of the factorization
	int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128);
	if(upper_only)
		t128 = t128 / 2 + 1;
	if(t128 >= 192)
		launch_gemm<128, 128, 64, 64, 0>(ctx->stream, m, n, (int)k, A, lda, B, ldb, C, ldc, upper_only);
	else
		launch_gemm<64, 64, 32, 32, 0>(ctx->stream, m, n, (int)k, A, lda, B, ldb, C, ldc, upper_only);
	SPP_HIP_CHECK(hipGetLastError());
}

// --------------------------------------------------------------------------------------------------
// potrf of one 128 x 128 diagonal block in LDS.
//   in : T = upper triangle of the block (global, column-major, ld)
//   out: upper triangle <- R_kk ; tinv (128 x 128, column-major, dense upper triangular) <- R_kk^-1
// Square-root-free right-looking elimination (rows are scaled once at the end), so a pivot step
// needs a single barrier. The strictly lower triangle of the LDS image accumulates G = (R^-1)^T:
//   step j, row i > j, f = T[j][i] / p_j:
//     c <  j : T[i][c] -= f * T[j][c]      (G update)
//     c == j : T[i][j]  = -f               (new G entry; G[j][j] = 1 unscaled)
//     c >= i : T[i][c] -= f * T[j][c]      (trailing update of the upper triangle)
// n_valid < 128 marks the last (padded) block: rows/cols >= n_valid are never pivots; column
// n_valid (if has_rhs) is carried along as a right-hand side.
// info[0] = first failing global pivot index + 1 (non-positive pivot, Eigen's LLT test).
// --------------------------------------------------------------------------------------------------
constexpr int NB = DENSE_NB;
constexpr int TS = NB + 1; // LDS row stride (column-major image: element (r, c) at r + c * TS)

__global__ __launch_bounds__(1024)
void potrf_diag_kernel(double *__restrict__ Ablk, int64_t ld, int n_valid, int has_rhs,
	double *__restrict__ tinv, int *__restrict__ info, int64_t k0)
{
	extern __shared__ double T[]; // NB * TS doubles + NB pivots
	double *piv = T + NB * TS;
	const int tid = threadIdx.x;
	for(int e = tid; e < NB * NB; e += 1024) {
		int r = e & (NB - 1), c = e >> 7;
		T[r + c * TS] = Ablk[r + (int64_t)c * ld];
	}
	__syncthreads();
	const int ncol = has_rhs ? (n_valid + 1 < NB ? n_valid + 1 : NB) : n_valid; // columns carried along
	// each thread owns rows i = tid >> 3 (+128 k... only 128 rows), column lanes (tid & 7) + 8 t
	const int i = tid >> 3, cl = tid & 7;
	bool failed = false;
	for(int j = 0; j < n_valid; ++ j) {
		const double p = T[j + j * TS];
		if(!(p > 0)) {
			failed = true; // uniform: every thread reads the same pivot
			if(tid == 0)
				info[0] = (int)(k0 + j + 1);
			break;
		}
		if(i > j && i < n_valid) {
			const double f = T[j + i * TS] / p; // R[j][i] / p_j (unscaled row j)
			for(int c = cl; c < j; c += 8)
				T[i + c * TS] -= f * T[j + c * TS];
			if(cl == (j & 7))
				T[i + j * TS] = -f;
			for(int c = i + ((cl - i) & 7); c < ncol; c += 8)
				T[i + c * TS] -= f * T[j + c * TS];
		}
		__syncthreads();
	}
	if(failed)
		return;
	if(tid < NB)
		piv[tid] = (tid < n_valid) ? 1.0 / sqrt(T[tid + tid * TS]) : 1.0;
	__syncthreads();
	// scale: R[j][c] = T[j][c] * piv[j] (c >= j); G[i][c] = T[i][c] * piv[i] (c < i), G[i][i] = piv[i]
	// write R (upper triangle + carried columns) back and the dense upper-triangular inverse
	for(int e = tid; e < NB * NB; e += 1024) {
		int r = e & (NB - 1), c = e >> 7;
		if(r < n_valid && c >= r && c < ncol)
			Ablk[r + (int64_t)c * ld] = (c == r) ? 1.0 / piv[r] : T[r + c * TS] * piv[r];
		// tinv[r][c] (upper, r <= c) = G[c][r]
		double v = 0;
		if(r == c)
			v = piv[r];
		else if(r < c && c < n_valid)
			v = T[c + r * TS] * piv[c];
		tinv[r + c * NB] = v;
	}
}

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
		const int r = tid & (NB - 1), h = tid >> 7;
		double s = 0;
		for(int c = h * 64; c < h * 64 + 64; ++ c)
			if(c < nv) s += tinv[r + c * NB] * y[k0 + c];
		part[h][r] = s;
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
		double s = 0;
#pragma unroll 8
		for(int c = 0; c < nv; ++ c) // real columns only: the padding of the last block holds the rhs column
			s += row[(int64_t)c * ld] * xk[c];
		y[i] -= s;
	}
}

__global__ void set_info_kernel(int *info) { info[0] = 0; }

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
static void ensure_dense_work(spp_ctx *ctx, int64_t nblk)
{
	ctx->dense.info.reserve(4);
	ctx->dense.tinv_all.reserve((size_t)nblk * NB * NB);
	ctx->dense.xtmp.reserve((size_t)(nblk + 1) * NB);
	static bool attr_set = false;
	if(!attr_set) {
		SPP_HIP_CHECK(hipFuncSetAttribute((const void*)potrf_diag_kernel,
			hipFuncAttributeMaxDynamicSharedMemorySize, (NB * TS + NB) * (int)sizeof(double)));
		attr_set = true;
	}
}

// Factor the n_pad x n_pad matrix in d_A (ld = n_pad multiple of 128, n < n_pad real columns; column
// n is carried along as right-hand side when has a padding column). Returns SPP_OK / SPP_NOT_POSDEF.
int dense_potrf_upper(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld, bool /*keep_inverses*/)
{
	SPP_REQUIRE(ld % NB == 0 && n < ld, SPP_E_BADARG, "dense_potrf_upper: ld must be a multiple of 128 and > n");
	const int64_t nblk = (n + NB - 1) / NB;
	ensure_dense_work(ctx, nblk);
	hipStream_t s = ctx->stream;
	hipLaunchKernelGGL(set_info_kernel, dim3(1), dim3(1), 0, s, ctx->dense.info.p);
	const int64_t ncols = n + 1; // real columns + rhs column
	for(int64_t k = 0; k < nblk; ++ k) {
		const int64_t k0 = k * NB;
		const int n_valid = (int)((n - k0 < NB) ? (n - k0) : NB);
		double *tinv = ctx->dense.tinv_all.p + (size_t)k * NB * NB;
		hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(1024), (NB * TS + NB) * sizeof(double), s,
			d_A + k0 + k0 * ld, ld, n_valid, (n_valid < NB) ? 1 : 0, tinv, ctx->dense.info.p, k0);
		const int64_t c1 = k0 + NB; // first column right of the block
		if(c1 >= ncols)
			break;
		const int64_t mrest = ncols - c1;
		// panel: R_kj = Tinv^T S_kj, in place (A = tinv: K x M = 128 x 128; B = C = S[k0.., c1..])
		launch_gemm<128, 32, 32, 32, 1>(s, NB, mrest, NB, tinv, NB, d_A + k0 + c1 * ld, ld,
			d_A + k0 + c1 * ld, ld, false);
		// trailing update: S[c1.., c1..] -= P^T P (upper tiles only), P = S[k0..k0+128, c1..]
		const int64_t mrows = ((n < ld ? n : ld) - c1); // rows that matter: real rows only
		if(mrows > 0) {
			dom_begin(ctx);
			dense_gemm_tn_sub(ctx, mrows, mrest, NB, d_A + k0 + c1 * ld, ld, d_A + k0 + c1 * ld, ld,
				d_A + c1 + c1 * ld, ld, true);
			// flops actually useful: upper triangle incl. rhs column
			dom_end(ctx, 2.0 * NB * (0.5 * (double)mrows * (double)mrows + (double)mrows));
		}
	}
	SPP_HIP_CHECK(hipGetLastError());
	int h_info = 0;
	SPP_HIP_CHECK(hipMemcpyAsync(&h_info, ctx->dense.info.p, sizeof(int), hipMemcpyDeviceToHost, s));
	SPP_HIP_CHECK(hipStreamSynchronize(s));
	return h_info ? SPP_NOT_POSDEF : SPP_OK;
}

// back substitution R x = y with y in d_b (n entries); uses the block inverses of the last potrf.
// d_b may be the rhs column of the factored matrix itself.
void dense_potrs_upper(spp_ctx *ctx, const double *d_R, int64_t n, int64_t ld, double *d_b)
{
	const int64_t nblk = (n + NB - 1) / NB;
	hipStream_t s = ctx->stream;
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

__global__ __launch_bounds__(256)
void mfma_f64_peak_kernel(double *out, int iters)
{
	v4f64 acc[8];
#pragma unroll
	for(int i = 0; i < 8; ++ i)
		acc[i] = (v4f64){0, 0, 0, 0};
	double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
	for(int it = 0; it < iters; ++ it) {
#pragma unroll
		for(int i = 0; i < 8; ++ i)
			acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	double s = 0;
#pragma unroll
	for(int i = 0; i < 8; ++ i)
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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
	// per wave: iters * 8 MFMAs of 16*16*4*2 flops
	double flops = (double)nblk * 4 * (double)iters * 8 * 2048.0;
	return flops / (ms * 1e-3) * 1e-12;
}

} // namespace spp
