// spp_schur.hip -- landmark elimination (Schur complement) on gfx950, fp64.
//
// Replaces the numeric part of CLinearSolver_Schur::Solve_PosDef_Blocky
// (reference include/slam/LinearSolver_Schur.h:1623-1935), phase by phase:
//   cinv_kernel      C^-1 block by block, negated        :1721-1735  (InverseOf_BlockDiag_FBS_Parallel,
//                                                         BlockMatrixFBS.inl:1750-1868; Eigen cofactor
//                                                         inverse BlockMatrixBase.h:1257-1270; Scale(-1))
//   obs_kernel       W = U * (-C^-1), W l                 :1737-1745 (MultiplyToWith_FBS #1), :1829
//   s_accum_kernel   S = W * U^T (upper) + A              :1757-1767 (MultiplyToWith_FBS #2 + AddTo_FBS)
//   rhs_kernel       x = eta_pose + sum W l               :1811-1830 (PreMultiply_Add_FBS)
//   [dense LLT, spp_dense.hip]                            :1839-1853
//   backsubst_obs/lm l = -l + U^T dx ; dl = (-C^-1) l     :1867-1881 (PostMultiply_Add_FBS_Parallel, PreMultiply_Add)
// The reference's Permute_UpperTriangular_To / SliceTo / TransposeTo (:1688-1709) move no data here:
// the kernels address the blocks of the ORIGINAL Lambda through the index lists built once by
// build_schur_plan() (spp_symbolic.cpp).
//
// Accumulation order of S: the pair list of every S block is sorted by landmark, which is the
// order in which the reference's product walks the columns of V; sums are sequential per output
// element (no atomics) => bit-reproducible run to run, and order-identical to the reference for
// blocks that are not split into chunks.
//
// Roofline: all kernels here are HBM/L2-bandwidth bound (0.25-1.5 flop/B); nothing is
// reshaped into GEMMs.

#include "spp_internal.h"

namespace spp {

// ---- -(C^-1) of one landmark block (m: DL x DL column-major) ------------------------------------------
template <int DL>
__device__ __forceinline__ void cinv_block(const double *__restrict__ m, double *o)
{
	if(DL == 3) {
		// cofactor formula with ONE reciprocal, determinant expanded along column 0
		// (what Eigen's fixed-size 3x3 inverse evaluates)
#define M_(i, j) m[(i) + 3 * (j)]
#define COF_(i, j) (M_(((i) + 1) % 3, ((j) + 1) % 3) * M_(((i) + 2) % 3, ((j) + 2) % 3) - \
	M_(((i) + 1) % 3, ((j) + 2) % 3) * M_(((i) + 2) % 3, ((j) + 1) % 3))
		const double c00 = COF_(0, 0), c10 = COF_(1, 0), c20 = COF_(2, 0);
		const double det = c00 * M_(0, 0) + c10 * M_(1, 0) + c20 * M_(2, 0);
		const double id = -1.0 / det; // negated: Scale(-1), LinearSolver_Schur.h:1735
		o[0] = c00 * id; o[3] = c10 * id; o[6] = c20 * id;
		o[1] = COF_(0, 1) * id; o[4] = COF_(1, 1) * id; o[7] = COF_(2, 1) * id;
		o[2] = COF_(0, 2) * id; o[5] = COF_(1, 2) * id; o[8] = COF_(2, 2) * id;
#undef COF_
#undef M_
	} else if(DL == 2) {
		const double det = m[0] * m[3] - m[2] * m[1];
		const double id = -1.0 / det;
		o[0] = m[3] * id; o[2] = -m[2] * id;
		o[1] = -m[1] * id; o[3] = m[0] * id;
	} else {
		// 6 x 6 (MIS cut of a 3D pose graph): in-register Gauss-Jordan on [M | I] without pivoting (M is a
		// diagonal block of an SPD matrix); Eigen's general inverse there is an LU with partial pivoting --
		// same result to rounding on SPD blocks
		double a[DL][DL], b[DL][DL];
#pragma unroll
		for(int i = 0; i < DL; ++ i)
#pragma unroll
			for(int j = 0; j < DL; ++ j) {
				a[i][j] = m[i + DL * j];
				b[i][j] = (i == j) ? 1.0 : 0.0;
			}
#pragma unroll
		for(int k = 0; k < DL; ++ k) {
			const double ip = 1.0 / a[k][k];
#pragma unroll
			for(int j = 0; j < DL; ++ j) {
				a[k][j] *= ip;
				b[k][j] *= ip;
			}
#pragma unroll
			for(int i = 0; i < DL; ++ i) {
				if(i == k)
					continue;
				const double f = a[i][k];
#pragma unroll
				for(int j = 0; j < DL; ++ j) {
					a[i][j] -= f * a[k][j];
					b[i][j] -= f * b[k][j];
				}
			}
		}
#pragma unroll
		for(int i = 0; i < DL; ++ i)
#pragma unroll
			for(int j = 0; j < DL; ++ j)
				o[i + DL * j] = -b[i][j];
	}
}

// factored operand of the S accumulation: C = G G^T (Cholesky of the block itself, not of its computed inverse),
// F = G^-T (upper triangular), C^-1 = F F^T -- so that W U^T = -(U F)(U F)^T needs ONE packed block per observation.
// F[t + DL * q] = Ginv[q][t], zero for t > q
template <int DL>
__device__ __forceinline__ void lfac_block(const double *__restrict__ m, double *F)
{
	double G[DL][DL], Gi[DL][DL];
#pragma unroll
	for(int j = 0; j < DL; ++ j)
#pragma unroll
		for(int i = 0; i < DL; ++ i) {
			G[i][j] = 0;
			if(i < j)
				continue;
			double sum = m[i + DL * j];
#pragma unroll
			for(int t = 0; t < DL; ++ t)
				if(t < j)
					sum -= G[i][t] * G[j][t];
			G[i][j] = (i == j) ? sqrt(sum) : sum / G[j][j];
		}
#pragma unroll
	for(int j = 0; j < DL; ++ j)
#pragma unroll
		for(int i = 0; i < DL; ++ i) {
			Gi[i][j] = 0;
			if(i < j)
				continue;
			if(i == j) {
				Gi[i][j] = 1.0 / G[j][j];
				continue;
			}
			double sum = 0;
#pragma unroll
			for(int t = 0; t < DL; ++ t)
				if(t >= j && t < i)
					sum += G[i][t] * Gi[t][j];
			Gi[i][j] = -sum / G[i][i];
		}
#pragma unroll
	for(int q = 0; q < DL; ++ q)
#pragma unroll
		for(int t = 0; t < DL; ++ t)
			F[t + DL * q] = Gi[q][t];
}

// ---- -(C^-1) (and F), one thread per landmark. With the factored accumulation and small landmark blocks (DL <= 3) this
// kernel is NOT launched: obs_fact_kernel forms F and backsubst_lm_kernel forms -(C^-1) from the block itself, where they
// are used (72 bytes read either way; 144 bytes per landmark written and read back, and a launch, less)
template <int DL>
__global__ __launch_bounds__(256)
void cinv_kernel(int64_t nl, const int64_t *__restrict__ lm_coff, const double *__restrict__ vals,
	double *__restrict__ cinv, double *__restrict__ lfac)
{
	const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(l >= nl)
		return;
	const double *m = vals + lm_coff[l];
	double o[DL * DL];
	cinv_block<DL>(m, o);
#pragma unroll
	for(int e = 0; e < DL * DL; ++ e)
		cinv[l * DL * DL + e] = o[e];
	if(lfac) {
		double F[DL * DL];
		lfac_block<DL>(m, F);
#pragma unroll
		for(int e = 0; e < DL * DL; ++ e)
			lfac[l * DL * DL + e] = F[e];
	}
}

// pose-landmark block of Lambda as DP x DL column-major; (offset << 1) | transposed flag
template <int DP, int DL>
__device__ __forceinline__ void load_U(const double *__restrict__ vals, int64_t oo, double *U)
{
	const double *src = vals + (oo >> 1);
	if(oo & 1) { // stored transposed (landmark id < pose id): DL x DP column-major
#pragma unroll
		for(int r = 0; r < DP; ++ r)
#pragma unroll
			for(int q = 0; q < DL; ++ q)
				U[r + DP * q] = src[q + DL * r];
	} else {
#pragma unroll
		for(int e = 0; e < DP * DL; ++ e)
			U[e] = src[e];
	}
}

// ---- W = U (-C^-1), packed U, W l : one lane per observation ----------------------------------------
// The camera-major slots of a wave's observations are scattered: with a lane storing its own block every
// store instruction put 64 eight-byte pieces into 64 different lines (36 + 6 such instructions per wave).
// W, the packed U and W l instead go through an LDS image [observation][odd stride] and are stored with
// consecutive lanes on consecutive doubles of one block (a few 64-byte segments per instruction).
template <int DP, int DL>
__global__ __launch_bounds__(256)
void obs_kernel(int64_t no, const int32_t *__restrict__ obs_lm, const int64_t *__restrict__ obs_off,
	const int64_t *__restrict__ lm_rbase, const double *__restrict__ vals, const double *__restrict__ rhs,
	const double *__restrict__ cinv, const int32_t *__restrict__ obs_wpos, double *__restrict__ W,
	double *__restrict__ Up, double *__restrict__ xw, int u_lm)
{
	constexpr int BLK = DP * DL, ST = BLK | 1;
	__shared__ double img_all[4][64 * ST];
	__shared__ int32_t slot_all[4][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double *img = img_all[wave];
	int32_t *slot = slot_all[wave];
	const int64_t a0 = ((int64_t)blockIdx.x * 4 + wave) * 64, a = a0 + lane;
	const bool active = a < no;
	const int nact = (no - a0 < 64) ? (int)((no - a0 > 0) ? (no - a0) : 0) : 64;
	const int64_t oo = active ? obs_off[a] : 0;
	const int32_t l = active ? obs_lm[a] : 0;
	slot[lane] = active ? obs_wpos[a] : 0; // camera-major slot of this observation
	double U[BLK];
	if(active)
		load_U<DP, DL>(vals, oo, U);
	else {
#pragma unroll
		for(int e = 0; e < BLK; ++ e)
			U[e] = 0;
	}
	double Ci[DL * DL];
#pragma unroll
	for(int e = 0; e < DL * DL; ++ e)
		Ci[e] = cinv[(int64_t)l * DL * DL + e];
	double lv[DL];
#pragma unroll
	for(int q = 0; q < DL; ++ q)
		lv[q] = active ? rhs[lm_rbase[l] + q] : 0.0;
	double Wl[DP], Wv[BLK];
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		Wl[r] = 0;
#pragma unroll
	for(int q = 0; q < DL; ++ q)
#pragma unroll
		for(int r = 0; r < DP; ++ r) {
			double sum = 0;
#pragma unroll
			for(int t = 0; t < DL; ++ t)
				sum += U[r + DP * t] * Ci[t + DL * q];
			Wv[r + DP * q] = sum;
			Wl[r] += sum * lv[q];
		}
	// block-cooperative stores: element p of the wave's 64 x BLK image goes to slot[p / BLK] * BLK + p % BLK
#pragma unroll
	for(int e = 0; e < BLK; ++ e)
		img[lane * ST + e] = Wv[e];
	__syncthreads();
	for(int p = lane; p < nact * BLK; p += 64) {
		const int j = p / BLK, e = p - j * BLK;
		W[(int64_t)slot[j] * BLK + e] = img[j * ST + e];
	}
	__syncthreads();
#pragma unroll
	for(int e = 0; e < BLK; ++ e)
		img[lane * ST + e] = U[e];
	__syncthreads();
	for(int p = lane; p < nact * BLK; p += 64) {
		const int j = p / BLK, e = p - j * BLK;
		Up[(u_lm ? a0 + j : (int64_t)slot[j]) * BLK + e] = img[j * ST + e]; // landmark-major = observation order
	}
	__syncthreads();
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		img[lane * (DP | 1) + r] = Wl[r];
	__syncthreads();
	for(int p = lane; p < nact * DP; p += 64) {
		const int j = p / DP, r = p - j * DP;
		xw[(int64_t)slot[j] * DP + r] = img[j * (DP | 1) + r];
	}
}

// ---- factored form: V = U F (C^-1 = F F^T, F = chol(C)^-T), V t with t = F^T l : one lane per observation ---------------------------
// One packed block per observation, in OBSERVATION order (landmark-major: the observers of a landmark are one
// contiguous run, so the two operands of a block product V_a V_b^T lie in the same few cache lines). A DP x DL = 6 x 3
// block is stored as 16 doubles in a 128-byte-aligned row of Vm + 2 doubles in the side array Vs: a gather pulls ONE
// line per block (+ a 16-byte piece of a line that eight neighbouring observations share) instead of the two or three
// lines a 144-byte block straddles. xw = W l = -V (F^T l) stays camera-major (rhs_kernel sums a pose's contiguous list).
template <int BLK> struct VSplit { static constexpr int IL = (BLK == 18) ? 16 : BLK, SIDE = BLK - IL; };

template <int DP, int DL>
__global__ __launch_bounds__(256)
void obs_fact_kernel(int64_t no, const int32_t *__restrict__ obs_lm, const int64_t *__restrict__ obs_off,
	const int64_t *__restrict__ lm_rbase, const double *__restrict__ vals, const double *__restrict__ rhs,
	const double *__restrict__ lfac, const int64_t *__restrict__ lm_coff, const int32_t *__restrict__ obs_wpos, double *__restrict__ Vm, double *__restrict__ Vs,
	double *__restrict__ xw)
{
	constexpr int BLK = DP * DL, ST = BLK | 1, IL = VSplit<BLK>::IL, SIDE = VSplit<BLK>::SIDE;
	__shared__ double img_all[4][64 * ST];
	__shared__ int32_t slot_all[4][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double *img = img_all[wave];
	int32_t *slot = slot_all[wave];
	const int64_t a0 = ((int64_t)blockIdx.x * 4 + wave) * 64, a = a0 + lane;
	const bool active = a < no;
	const int nact = (no - a0 < 64) ? (int)((no - a0 > 0) ? (no - a0) : 0) : 64;
	const int64_t oo = active ? obs_off[a] : 0;
	const int32_t l = active ? obs_lm[a] : 0;
	slot[lane] = active ? obs_wpos[a] : 0; // camera-major slot of this observation (xw)
	double U[BLK];
	if(active)
		load_U<DP, DL>(vals, oo, U);
	else {
#pragma unroll
		for(int e = 0; e < BLK; ++ e)
			U[e] = 0;
	}
	double F[DL * DL], lv[DL], tv[DL];
	if(lfac) {
#pragma unroll
		for(int e = 0; e < DL * DL; ++ e)
			F[e] = lfac[(int64_t)l * DL * DL + e];
	} else
		lfac_block<DL>(vals + lm_coff[l], F); // (the observers of a landmark are neighbouring lanes: its block comes out of the cache)
#pragma unroll
	for(int q = 0; q < DL; ++ q)
		lv[q] = active ? rhs[lm_rbase[l] + q] : 0.0;
#pragma unroll
	for(int q = 0; q < DL; ++ q) { // t = F^T l (F upper: t_q = sum_{u <= q} F[u][q] l_u)
		double sum = 0;
#pragma unroll
		for(int u = 0; u < DL; ++ u)
			if(u <= q)
				sum += F[u + DL * q] * lv[u];
		tv[q] = sum;
	}
	double V[BLK], Wl[DP];
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		Wl[r] = 0;
#pragma unroll
	for(int q = 0; q < DL; ++ q)
#pragma unroll
		for(int r = 0; r < DP; ++ r) {
			double sum = 0;
#pragma unroll
			for(int t = 0; t < DL; ++ t)
				if(t <= q)
					sum += U[r + DP * t] * F[t + DL * q];
			V[r + DP * q] = sum;
			Wl[r] -= sum * tv[q];
		}
#pragma unroll
	for(int e = 0; e < BLK; ++ e)
		img[lane * ST + e] = V[e];
	__syncthreads();
	for(int p = lane; p < nact * IL; p += 64) { // the in-line parts of the wave's 64 blocks are one contiguous range
		const int j = p / IL, e = p - j * IL;
		Vm[a0 * IL + p] = img[j * ST + e];
	}
	if(SIDE)
		for(int p = lane; p < nact * SIDE; p += 64) {
			const int j = p / (SIDE ? SIDE : 1), e = p - j * SIDE;
			Vs[a0 * SIDE + p] = img[j * ST + IL + e];
		}
	__syncthreads();
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		img[lane * (DP | 1) + r] = Wl[r];
	__syncthreads();
	for(int p = lane; p < nact * DP; p += 64) {
		const int j = p / DP, r = p - j * DP;
		xw[(int64_t)slot[j] * DP + r] = img[j * (DP | 1) + r];
	}
}

// ---- S block accumulation: a wave works through a short run of work items ------------------------------
// Each LANE takes whole pairs (a, b) of the item's list (lane, lane + 64, ...) and forms the DP x DP
// outer product sum W_a U_b^T in registers (108 FMAs per pair). The blocks are NOT fetched by the lane
// that consumes them: a lane-per-block gather makes every load instruction touch 64 different cache
// lines and the kernel was bound by the CU's address/tag path at 13 % VALU utilization. Instead the
// wave fetches the blocks cooperatively -- DP*DL/2 consecutive lanes read the 16-byte pieces of one
// block, 7 blocks (14 lines) per instruction -- into an LDS image [pair][19 doubles] (the odd stride
// makes the per-lane ds_read_b64 of the consumer conflict-free), from which every lane then reads its
// own pair. The 64 partial blocks are summed IN LANE ORDER through LDS, so for lists of up to 64 pairs
// (the common case) the additions happen in exactly the order of the pair list (= ascending landmark
// = the reference's order); longer lists add lane-strided partial sums. No atomics: bit-reproducible.
//
// What bounds it (rocprofv3 counters, Venice shape, profiles/r02_venice871_s_accum_pmc.txt): the SIMDs' issue
// ports. A wave lives ~15 500 cycles, a third of them issuing (640 vector + 137 LDS + 36 vector-memory
// instructions per item of ~70 pairs), a third stalled behind the other wave of its SIMD, a third waiting for
// memory; 8 waves per CU (LDS images + 232 VGPRs). The time did not follow the L2 misses (2.6e7 ... 4.0e7
// 128-byte lines per launch across item orders: 0.85 ... 0.98 ms) nor a software pipeline across items.
// Hence: (i) an item is ONE 32-byte record (SaccItem: pair range, destination, A offset) instead of four
// dependent index loads, (ii) the block indices of a round are shuffled in one batch and every gather load is
// unconditional (no branch per load), (iii) the reduction walks three segments of the partial list side by side
// with 16-byte LDS reads. A wave can own `chunk` items, holding the pair indices of the NEXT round, the record
// after that and the A block while a gather is in flight; with one item per wave (the default) that machinery
// idles.
constexpr int SACC_WAVES = 4; // waves per workgroup
#ifndef SPP_SACC_RP
#define SPP_SACC_RP 64 // pairs per round
#endif
#ifndef SPP_SACC_UCOL
#define SPP_SACC_UCOL 0
#endif
#ifndef SPP_SACC_GLDS
#define SPP_SACC_GLDS 1 // 1: the gathered blocks go from L2 / HBM straight into the LDS images (global_load_lds_dwordx4)
#endif

__device__ __forceinline__ SaccItem sacc_load_item(const SaccItem *__restrict__ items, int32_t idx, int32_t lim)
{
	SaccItem r;
	r.beg = r.end = 0;
	r.kind = -1;
	r.pad = 0;
	r.dst = 0;
	r.aoff = -1;
	if(idx < lim)
		r = items[idx];
	return r;
}

#ifndef SPP_SACC_EU
#define SPP_SACC_EU 2 // waves per SIMD the register allocator leaves room for (2: 256 VGPRs; the LDS images allow 8 waves per CU anyway)
#endif
// FACT: both operands of a product are packed blocks V of the factored form (obs_fact_kernel): W = the in-line array Vm,
// Up = the side array Vs, pair_a / pair_b both observation indices; the accumulators collect -V_a V_b^T
template <int DP, int DL, bool FACT>
__global__ __launch_bounds__(SACC_WAVES * 64) __attribute__((amdgpu_waves_per_eu(SPP_SACC_EU, SPP_SACC_EU)))
void s_accum_kernel(const SaccItem *__restrict__ items, const int32_t *__restrict__ xcd_beg, int chunk,
	const int32_t *__restrict__ pair_a, const int32_t *__restrict__ pair_b,
	const double *__restrict__ W, const double *__restrict__ Up, const double *__restrict__ vals,
	double *__restrict__ S, int64_t ld, double *__restrict__ partial)
{
	constexpr int NE = DP * DP, BLK = DP * DL;
	constexpr int IL = VSplit<BLK>::IL, SIDE = VSplit<BLK>::SIDE;
	constexpr int PW = (BLK & 1) ? 1 : 2;       // doubles per fetched piece: 16-byte pieces need an even block (9-double
	                                            // blocks of the 3 x 3 case are only 8-byte aligned: 8-byte pieces)
	constexpr int PCS = BLK / PW;               // pieces per block
	constexpr int PPI = 64 / PCS;               // blocks (pairs) fetched per wave instruction
	constexpr int RP = SPP_SACC_RP;             // pairs per round: the LDS images hold RP blocks per operand
	constexpr int NG = (RP + PPI - 1) / PPI;    // instructions per operand and round
	// GLDS: the blocks are fetched by LDS-DMA: no staging registers, no ds_write -- the LDS pipe was the busiest unit of
	// the kernel (137 LDS instructions per item, a third of the issue cycles). One instruction lands the 16-byte pieces
	// of its 64 lanes back to back, i.e. PPI whole blocks at their natural stride of BLK doubles: the image is
	// [pair][BLK], which the per-lane ds_read_b128 of the consumer reads without bank conflicts (36 l mod 64 is
	// distinct over 16 lanes). The lanes behind the last whole block of an instruction stay masked (their 16 bytes
	// would land on the next group's first block).
	constexpr bool GLDS = (PW == 2) && (SPP_SACC_GLDS != 0);
	constexpr int ST = GLDS ? BLK : (BLK | 1);  // LDS stride of one block image in doubles (odd when staged through registers)
	constexpr int NE2 = (NE + 1) / 2;           // element pairs of a block
	constexpr int RS = 2 * NE2 + 2;             // row stride of the reduction image (even: 16-byte rows)
	constexpr int NSEG = 64 / NE2 < 3 ? 64 / NE2 : 3; // segments of the partial list summed side by side
	constexpr int LW = (2 * RP * ST > RP * RS) ? 2 * RP * ST : RP * RS; // per wave: the staging area, reused by the reduction image
	__shared__ __attribute__((aligned(16))) double lds[SACC_WAVES][LW];
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	// workgroups go round-robin over the 8 XCDs: XCD x works through its own contiguous range of the
	// items (equal work per range), so that the camera segments of a tile of blocks are fetched into
	// ONE L2 (gridDim.x is a multiple of 8). A workgroup owns SACC_WAVES * chunk consecutive items of that
	// range, wave w every SACC_WAVES-th of them.
	const int x = blockIdx.x & 7;
	const int32_t lim = xcd_beg[x + 1];
	int32_t idx = xcd_beg[x] + (int32_t)(blockIdx.x >> 3) * SACC_WAVES * chunk + wave;
	if(idx >= lim)
		return; // whole wave
	int left = chunk; // items this wave may still take
	double *sw = lds[wave], *su = sw + RP * ST, *rw = sw;
	const int my_pair = lane / PCS, my_piece = lane % PCS; // role in the cooperative fetch (lanes >= PPI * PCS idle)
	SaccItem cur = sacc_load_item(items, idx, lim);
	SaccItem nxt = sacc_load_item(items, (left > 1) ? idx + SACC_WAVES : lim, lim);
	SaccItem nn = nxt;
	int32_t q0 = cur.beg;
	int32_t pa = -1, pb = -1;
	if(lane < RP && q0 + lane < cur.end) {
		pa = pair_a[q0 + lane];
		pb = pair_b[q0 + lane];
	}
	double acc[NE];
#pragma unroll
	for(int e = 0; e < NE; ++ e)
		acc[e] = 0;
	for(;;) {
		const int nround = (cur.end - q0 < RP) ? cur.end - q0 : RP; // 0 for an item without pairs (A only)
		const bool mine = lane < nround; // this lane owns a pair of the round
		const bool last = q0 + RP >= cur.end; // last round of the item (wave-uniform)
		// ---- ahead of this round's gather: the pair indices of the next round, and at the end of an item the
		// record after the next one and this item's A block
		const int32_t nq0 = last ? nxt.beg : q0 + RP, nend = last ? nxt.end : cur.end;
		int32_t pa_n = -1, pb_n = -1;
		if(lane < RP && nq0 + lane < nend) {
			pa_n = pair_a[nq0 + lane];
			pb_n = pair_b[nq0 + lane];
		}
		double aval0 = 0, aval1 = 0;
		if(last) {
			nn = sacc_load_item(items, (left > 2) ? idx + 2 * SACC_WAVES : lim, lim);
			if(cur.aoff >= 0 && lane < NE2) {
				aval0 = vals[cur.aoff + 2 * lane];
				if(2 * lane + 1 < NE)
					aval1 = vals[cur.aoff + 2 * lane + 1];
			}
		}
		// ---- cooperative fetch of up to 64 W and 64 U blocks into the LDS images
		if constexpr(GLDS) {
			int32_t ia[NG], ib[NG];
#pragma unroll
			for(int g = 0; g < NG; ++ g) {
				const int p = g * PPI + my_pair; // pair of this round served by this lane
				ia[g] = __shfl(pa, p & 63);
				ib[g] = __shfl(pb, p & 63);
			}
#pragma unroll
			for(int g = 0; g < NG; ++ g) {
				const int p = g * PPI + my_pair;
				if(my_pair < PPI && p < nround) {
					const double *srca, *srcb;
					if(FACT) { // piece of the in-line row, or of the side array (a lane's piece index is fixed: no divergence inside a block)
						const bool side = SIDE && 2 * my_piece >= IL;
						srca = side ? Up + (int64_t)ia[g] * SIDE + (2 * my_piece - IL) : W + (int64_t)ia[g] * IL + 2 * my_piece;
						srcb = side ? Up + (int64_t)ib[g] * SIDE + (2 * my_piece - IL) : W + (int64_t)ib[g] * IL + 2 * my_piece;
					} else {
						srca = W + (int64_t)ia[g] * BLK + 2 * my_piece;
						srcb = Up + (int64_t)ib[g] * BLK + 2 * my_piece;
					}
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srca,
						(__attribute__((address_space(3))) void*)(sw + g * PPI * BLK), 16, 0, 0);
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcb,
						(__attribute__((address_space(3))) void*)(su + g * PPI * BLK), 16, 0, 0);
				}
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		} else {
		constexpr int GC = (NG < 10) ? NG : 10; // groups in flight at a time (register budget of the 6 x 6 case)
#pragma unroll
		for(int g0 = 0; g0 < NG; g0 += GC) {
			// block indices first -- one ds_bpermute each, all in flight together --, then the loads, none of them
			// behind a branch: a lane without a pair to serve (short round, the idle lanes behind PPI * PCS) fetches
			// from block 0 and its piece of the image is never read
			int32_t ia[GC], ib[GC];
#pragma unroll
			for(int gg = 0; gg < GC; ++ gg) {
				const int p = (g0 + gg) * PPI + my_pair; // pair of this round served by this lane
				ia[gg] = __shfl(pa, p & 63);
				ib[gg] = __shfl(pb, p & 63);
			}
			double tw[GC][PW], tu[GC][PW];
#pragma unroll
			for(int gg = 0; gg < GC; ++ gg) {
				int32_t xa = ia[gg] < 0 ? 0 : ia[gg], xb = ib[gg] < 0 ? 0 : ib[gg];
				if(PW == 2) {
					const double2 a2 = *(const double2*)(W + (int64_t)xa * BLK + 2 * my_piece);
					const double2 b2 = *(const double2*)((FACT ? W : Up) + (int64_t)xb * BLK + 2 * my_piece);
					tw[gg][0] = a2.x; tw[gg][PW - 1] = a2.y;
					tu[gg][0] = b2.x; tu[gg][PW - 1] = b2.y;
				} else {
					tw[gg][0] = W[(int64_t)xa * BLK + my_piece];
					tu[gg][0] = (FACT ? W : Up)[(int64_t)xb * BLK + my_piece];
				}
			}
#pragma unroll
			for(int gg = 0; gg < GC; ++ gg) {
				const int g = g0 + gg;
				const int p = g * PPI + my_pair;
				if(g < NG && my_pair < PPI && p < RP) {
#pragma unroll
					for(int t = 0; t < PW; ++ t) {
						sw[p * ST + PW * my_piece + t] = tw[gg][t];
						su[p * ST + PW * my_piece + t] = tu[gg][t];
					}
				}
			}
		}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// ---- every lane: its own pair out of LDS
		if(mine) {
#if SPP_SACC_UCOL
			// U column by column (DL values live at a time instead of the whole block): the register budget of three waves per SIMD
			double w[BLK];
#pragma unroll
			for(int e = 0; e < BLK; ++ e)
				w[e] = sw[lane * ST + e];
#pragma unroll
			for(int c = 0; c < DP; ++ c) {
				double uc[DL];
#pragma unroll
				for(int t = 0; t < DL; ++ t)
					uc[t] = su[lane * ST + c + DP * t];
#pragma unroll
				for(int r = 0; r < DP; ++ r) {
					double sp = 0;
#pragma unroll
					for(int t = 0; t < DL; ++ t)
						sp += w[r + DP * t] * uc[t];
					acc[r + DP * c] += sp;
				}
			}
#else
			double w[BLK], u[BLK];
#pragma unroll
			for(int e = 0; e < BLK; ++ e) {
				w[e] = sw[lane * ST + e];
				u[e] = su[lane * ST + e];
			}
#pragma unroll
			for(int c = 0; c < DP; ++ c)
#pragma unroll
				for(int r = 0; r < DP; ++ r) {
					double sp = 0;
#pragma unroll
					for(int t = 0; t < DL; ++ t)
						sp += w[r + DP * t] * u[c + DP * t];
					acc[r + DP * c] += FACT ? -sp : sp;
				}
#endif
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if(last) {
			int nact = cur.end - cur.beg;
			if(nact > RP)
				nact = RP;
			// Reduction of the partial blocks through LDS. The lanes that own no pair hold exact zeros. Lane (e2, q)
			// adds elements 2 e2, 2 e2 + 1 of the q-th segment of the partial list in list order; the NSEG segment
			// sums are then added in order: a fixed summation order (bit-reproducible), with a dependent chain a
			// third as long as one lane walking the whole list.
			if(lane < RP) { // (the image holds RP rows; lanes beyond own no pair and hold zeros)
#pragma unroll
				for(int e = 0; e < NE; ++ e)
					rw[lane * RS + e] = acc[e];
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			const int seg_len = (nact + NSEG - 1) / NSEG;
			const int e2 = lane % NE2, sq = lane / NE2;
			double s0 = 0, s1 = 0;
			if(sq < NSEG) {
				const double *src = rw + (sq * seg_len) * RS + 2 * e2;
				const int len = (seg_len < RP - sq * seg_len) ? seg_len : RP - sq * seg_len; // rows [nact, RP) are zeros
				int l = 0;
				for(; l + 8 <= len; l += 8) {
					double2 t[8];
#pragma unroll
					for(int u = 0; u < 8; ++ u)
						t[u] = *(const double2*)(src + (l + u) * RS);
#pragma unroll
					for(int u = 0; u < 8; ++ u) {
						s0 += t[u].x;
						s1 += t[u].y;
					}
				}
				for(; l < len; ++ l) {
					const double2 t = *(const double2*)(src + l * RS);
					s0 += t.x;
					s1 += t.y;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			{
				const double g0 = s0, g1 = s1; // the segment sums as they are before any lane combines them
#pragma unroll
				for(int q = 1; q < NSEG; ++ q) {
					s0 += __shfl(g0, (lane + q * NE2) & 63); // only lanes < NE2 keep a meaningful result
					s1 += __shfl(g1, (lane + q * NE2) & 63);
				}
			}
			if(lane < NE2) {
				const int e = 2 * lane;
				const bool two = e + 1 < NE;
				double *dst;
				int64_t o0 = e, o1 = e + 1;
				if(cur.kind == 2)
					dst = partial + cur.dst;
				else {
					dst = S + cur.dst;
					if(cur.aoff >= 0) { // AddTo_FBS: S = A + W V
						s0 = aval0 + s0;
						s1 = aval1 + s1;
					}
					if(cur.kind == 0) { // dp x dp block of the dense S
						o0 = e % DP + (e / DP) * ld;
						o1 = (e + 1) % DP + ((e + 1) / DP) * ld;
					}
				}
				dst[o0] = s0;
				if(two)
					dst[o1] = s1;
			}
#pragma unroll
			for(int e = 0; e < NE; ++ e)
				acc[e] = 0;
			idx += SACC_WAVES;
			-- left;
			cur = nxt;
			nxt = nn;
			if(left == 0 || cur.kind < 0)
				break;
		}
		q0 = nq0;
		pa = pa_n;
		pb = pb_n;
	}
}

// blocks whose pair list was split: sum the partial slots in order (+ A)
template <int DP>
__global__ __launch_bounds__(256)
void s_multi_kernel(int64_t n_multi, const int32_t *__restrict__ multi_blk, const int32_t *__restrict__ multi_ptr,
	const int32_t *__restrict__ sblk_i1, const int32_t *__restrict__ sblk_i2, const int64_t *__restrict__ sblk_aoff,
	const double *__restrict__ partial, const double *__restrict__ vals, int add_A, double *__restrict__ S, int64_t ld,
	const int64_t *__restrict__ sblk_voff)
{
	const int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if(m >= n_multi || lane >= DP * DP)
		return;
	const int32_t b = multi_blk[m];
	double acc = 0;
	for(int32_t s = multi_ptr[m]; s < multi_ptr[m + 1]; ++ s)
		acc += partial[(int64_t)s * DP * DP + lane];
	const int64_t aoff = sblk_aoff[b];
	if(add_A && aoff >= 0)
		acc = vals[aoff + lane] + acc;
	if(sblk_voff) {
		S[sblk_voff[b] + lane] = acc;
		return;
	}
	const int r = lane % DP, c = lane / DP;
	S[((int64_t)sblk_i1[b] * DP + r) + ((int64_t)sblk_i2[b] * DP + c) * ld] = acc;
}

// ---- reduced right-hand side: one wave per pose; written into padding column n_red of S -----------
template <int DP>
__global__ __launch_bounds__(256)
void rhs_kernel(int64_t nc, const int32_t *__restrict__ cam_ptr, const int32_t *__restrict__ cam_obs,
	const int64_t *__restrict__ pose_rbase, const double *__restrict__ xw, const double *__restrict__ rhs,
	int add_eta, double *__restrict__ xcol)
{
	const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if(i >= nc)
		return;
	double s[DP];
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		s[r] = 0;
	for(int32_t q = cam_ptr[i] + lane; q < cam_ptr[i + 1]; q += 64) {
		const double *x = xw + (int64_t)q * DP; // xw is camera-major: the pose's list is contiguous
#pragma unroll
		for(int r = 0; r < DP; ++ r)
			s[r] += x[r];
	}
#pragma unroll
	for(int r = 0; r < DP; ++ r) {
#pragma unroll
		for(int off = 32; off > 0; off >>= 1)
			s[r] += __shfl_xor(s[r], off);
	}
	if(lane < DP) {
		double v = 0;
#pragma unroll
		for(int r = 0; r < DP; ++ r)
			if(lane == r)
				v = s[r];
		if(add_eta)
			v += rhs[pose_rbase[i] + lane];
		xcol[i * DP + lane] = v;
	}
}

// ---- back-substitution ---------------------------------------------------------------------------
// dl = (-C^-1) (-l + sum_a U_a^T dx_pose(a)), LinearSolver_Schur.h:1867-1881. Two launches: the products
// U_a^T dx, one lane per observation (uniform work, neighbouring lanes read neighbouring blocks of a
// landmark's column of Lambda), then one thread per landmark adds its (contiguous) products in
// observation order and applies -C^-1. tq (DL doubles per observation) reuses the W l buffer.
template <int DP, int DL>
__global__ __launch_bounds__(256)
void backsubst_obs_kernel(int64_t no, const int32_t *__restrict__ obs_pose, const int64_t *__restrict__ obs_off,
	const double *__restrict__ vals, const double *__restrict__ dx, double *__restrict__ tq)
{
	const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(a >= no)
		return;
	double u[DP * DL];
	load_U<DP, DL>(vals, obs_off[a], u);
	double dv[DP];
	const double *d = dx + (int64_t)obs_pose[a] * DP;
#pragma unroll
	for(int r = 0; r < DP; ++ r)
		dv[r] = d[r];
#pragma unroll
	for(int q = 0; q < DL; ++ q) {
		double sum = 0;
#pragma unroll
		for(int r = 0; r < DP; ++ r)
			sum += dv[r] * u[r + DP * q];
		tq[a * DL + q] = sum;
	}
}

template <int DL>
__global__ __launch_bounds__(256)
void backsubst_lm_kernel(int64_t nl, const int32_t *__restrict__ lm_ptr, const int64_t *__restrict__ lm_rbase,
	const double *__restrict__ tq, const double *__restrict__ cinv, const int64_t *__restrict__ lm_coff, const double *__restrict__ vals,
	double *__restrict__ rhs)
{
	const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(l >= nl)
		return;
	double t[DL];
	const int64_t rb = lm_rbase[l];
#pragma unroll
	for(int q = 0; q < DL; ++ q)
		t[q] = -rhs[rb + q]; // v_l = -v_l, LinearSolver_Schur.h:1867
	for(int32_t a = lm_ptr[l]; a < lm_ptr[l + 1]; ++ a) {
#pragma unroll
		for(int q = 0; q < DL; ++ q)
			t[q] += tq[(int64_t)a * DL + q];
	}
	double Ci[DL * DL];
	if(cinv) {
#pragma unroll
		for(int e = 0; e < DL * DL; ++ e)
			Ci[e] = cinv[l * DL * DL + e];
	} else
		cinv_block<DL>(vals + lm_coff[l], Ci);
#pragma unroll
	for(int q = 0; q < DL; ++ q) {
		double sum = 0;
#pragma unroll
		for(int u = 0; u < DL; ++ u)
			sum += Ci[q + DL * u] * t[u];
		rhs[rb + q] = sum;
	}
}

// Both steps in one launch: a workgroup takes a group of consecutive landmarks with at most 256 observations between them
// (bs_ptr, built with the plan), forms the products U_a^T dx one lane per observation into LDS, then one lane per landmark
// adds them in observation order and applies -C^-1 -- the same operations in the same order as the two kernels above,
// without the round trip of the products through memory (24 B per observation written and read back) and its launch.
template <int DP, int DL>
__global__ __launch_bounds__(256)
void backsubst_fused_kernel(const int32_t *__restrict__ bs_ptr, const int32_t *__restrict__ lm_ptr, const int32_t *__restrict__ obs_pose,
	const int64_t *__restrict__ obs_off, const int64_t *__restrict__ lm_rbase, const double *__restrict__ cinv,
	const int64_t *__restrict__ lm_coff, const double *__restrict__ vals, const double *__restrict__ dx, double *__restrict__ rhs)
{
	__shared__ double tq[256 * DL];
	const int32_t l0 = bs_ptr[blockIdx.x], l1 = bs_ptr[blockIdx.x + 1];
	const int32_t a0 = lm_ptr[l0], na = lm_ptr[l1] - a0;
	const int t = threadIdx.x;
	if(t < na) {
		const int64_t a = (int64_t)a0 + t;
		double u[DP * DL];
		load_U<DP, DL>(vals, obs_off[a], u);
		double dv[DP];
		const double *d = dx + (int64_t)obs_pose[a] * DP;
#pragma unroll
		for(int r = 0; r < DP; ++ r)
			dv[r] = d[r];
#pragma unroll
		for(int q = 0; q < DL; ++ q) {
			double sum = 0;
#pragma unroll
			for(int r = 0; r < DP; ++ r)
				sum += dv[r] * u[r + DP * q];
			tq[t * DL + q] = sum;
		}
	}
	__syncthreads();
	if(t >= l1 - l0)
		return;
	const int64_t l = (int64_t)l0 + t;
	double tv[DL];
	const int64_t rb = lm_rbase[l];
#pragma unroll
	for(int q = 0; q < DL; ++ q)
		tv[q] = -rhs[rb + q]; // v_l = -v_l, LinearSolver_Schur.h:1867
	for(int32_t a = lm_ptr[l] - a0, e = lm_ptr[l + 1] - a0; a < e; ++ a) {
#pragma unroll
		for(int q = 0; q < DL; ++ q)
			tv[q] += tq[a * DL + q];
	}
	double Ci[DL * DL];
	if(cinv) {
#pragma unroll
		for(int e = 0; e < DL * DL; ++ e)
			Ci[e] = cinv[l * DL * DL + e];
	} else
		cinv_block<DL>(vals + lm_coff[l], Ci);
#pragma unroll
	for(int q = 0; q < DL; ++ q) {
		double sum = 0;
#pragma unroll
		for(int u = 0; u < DL; ++ u)
			sum += Ci[q + DL * u] * tv[u];
		rhs[rb + q] = sum;
	}
}

template <int DP>
__global__ __launch_bounds__(256)
void scatter_dx_kernel(int64_t nc, const int64_t *__restrict__ pose_rbase, const double *__restrict__ dx,
	double *__restrict__ rhs)
{
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= nc * DP)
		return;
	rhs[pose_rbase[e / DP] + e % DP] = dx[e];
}

// --------------------------------------------------------------------------------------------------
template <int DP, int DL>
static void schur_form_t(spp_ctx *ctx, const double *d_vals, const double *d_rhs, double *S)
{
	SchurPlan &sp = ctx->schur;
	hipStream_t s = ctx->stream;
	const int64_t ld = sp.ld;
	const int64_t *voff = sp.sparse_S ? sp.sblk_voff.p : nullptr;
	double *xcol = sp.sparse_S ? S + sp.s_st.nvals : S + sp.n_red * ld; // reduced rhs
	phase_begin(ctx, SPP_PHASE_SCHUR_INV);
	// Every written block of S is ASSIGNED by the accumulation kernels. When the (dense, unsharded) reduced system lives in
	// the solver's own buffer (not a caller's: the split API hands S to an all-reduce) and has all of its upper blocks written, only the padding columns -- which carry the reduced rhs -- need
	// clearing: nothing below the diagonal is ever an operand (220 MB memset -> 1 MB on the Venice shape).
	if(S == sp.S.p && !sp.sparse_S && ctx->shard_world == 1 && sp.n_sblk == sp.nc * (sp.nc + 1) / 2 && sp.n_red == sp.nc * DP)
		SPP_HIP_CHECK(hipMemsetAsync(S + sp.n_red * ld, 0, (size_t)(schur_buffer_doubles(ctx) - sp.n_red * ld) * sizeof(double), s));
	else
		SPP_HIP_CHECK(hipMemsetAsync(S, 0, (size_t)schur_buffer_doubles(ctx) * sizeof(double), s));
	constexpr int VIL = VSplit<DP * DL>::IL;
	double *Vm = sp.W.p, *Vs = sp.W.p + sp.no * VIL; // factored form: in-line rows, then the side array, in the one buffer W
	const bool onfly = sp.factored && DL <= 3; // F and -(C^-1) formed where they are used: no cinv_kernel
	if(sp.nl && !onfly)
		hipLaunchKernelGGL((cinv_kernel<DL>), dim3((unsigned)((sp.nl + 255) / 256)), dim3(256), 0, s,
			sp.nl, sp.lm_coff.p, d_vals, sp.cinv.p, sp.factored ? sp.lfac.p : nullptr);
	if(sp.no && sp.factored)
		hipLaunchKernelGGL((obs_fact_kernel<DP, DL>), dim3((unsigned)((sp.no + 255) / 256)), dim3(256), 0, s,
			sp.no, sp.obs_lm.p, sp.obs_off.p, sp.lm_rbase.p, d_vals, d_rhs, onfly ? (const double*)nullptr : (const double*)sp.lfac.p, sp.lm_coff.p,
			sp.obs_wpos.p, Vm, Vs, sp.xw.p);
	else if(sp.no)
		hipLaunchKernelGGL((obs_kernel<DP, DL>), dim3((unsigned)((sp.no + 255) / 256)), dim3(256), 0, s,
			sp.no, sp.obs_lm.p, sp.obs_off.p, sp.lm_rbase.p, d_vals, d_rhs, sp.cinv.p, sp.obs_wpos.p, sp.W.p, sp.Up.p, sp.xw.p, sp.u_landmark_major ? 1 : 0);
	phase_end(ctx, SPP_PHASE_SCHUR_INV);
	phase_begin(ctx, SPP_PHASE_SCHUR_GEMM);
	if(sp.n_items) {
		static int chunk_env = -1;
		if(chunk_env < 0) {
			// items per wave (0 = one persistent set of workgroups). Measured on the Venice shape: 1 -> 0.86 ms, 4...16 ->
			// 0.98 ms, persistent 1.5 ms: the hardware's dynamic dispatch of one-item waves balances the uneven items
			// (1 ... 2048 pairs) better than the software pipeline across items hides latency
			const char *e = getenv("SPP_SACC_CHUNK");
			chunk_env = e ? atoi(e) : 1;
		}
		int chunk = chunk_env;
		if(chunk <= 0) // persistent: two workgroups per CU (the LDS images of 8 waves fill a CU)
			chunk = (int)((sp.xcd_max_items + SACC_WAVES * 64 - 1) / (SACC_WAVES * 64));
		const int64_t per = (int64_t)SACC_WAVES * chunk;
		if(sp.factored)
			hipLaunchKernelGGL((s_accum_kernel<DP, DL, true>), dim3((unsigned)(8 * ((sp.xcd_max_items + per - 1) / per))), dim3(SACC_WAVES * 64), 0, s,
				sp.items.p, sp.xcd_beg.p, chunk, sp.pair_a.p, sp.pair_b.p, Vm, Vs, d_vals, S, ld, sp.partial.p);
		else
			hipLaunchKernelGGL((s_accum_kernel<DP, DL, false>), dim3((unsigned)(8 * ((sp.xcd_max_items + per - 1) / per))), dim3(SACC_WAVES * 64), 0, s,
				sp.items.p, sp.xcd_beg.p, chunk, sp.pair_a.p, sp.pair_b.p, sp.W.p, sp.Up.p, d_vals, S, ld, sp.partial.p);
	}
	if(sp.n_multi)
		hipLaunchKernelGGL((s_multi_kernel<DP>), dim3((unsigned)((sp.n_multi + 3) / 4)), dim3(256), 0, s,
			sp.n_multi, sp.multi_blk.p, sp.multi_ptr.p, sp.sblk_i1.p, sp.sblk_i2.p, sp.sblk_aoff.p,
			sp.partial.p, d_vals, sp.add_A ? 1 : 0, S, ld, voff);
	phase_end(ctx, SPP_PHASE_SCHUR_GEMM);
	phase_begin(ctx, SPP_PHASE_SCHUR_RHS);
	if(sp.nc)
		hipLaunchKernelGGL((rhs_kernel<DP>), dim3((unsigned)((sp.nc + 3) / 4)), dim3(256), 0, s,
			sp.nc, sp.cam_ptr.p, sp.cam_obs.p, sp.pose_rbase.p, sp.xw.p, d_rhs, sp.add_A ? 1 : 0, xcol);
	phase_end(ctx, SPP_PHASE_SCHUR_RHS);
	SPP_HIP_CHECK(hipGetLastError());
}

template <int DP, int DL>
static int schur_finish_t(spp_ctx *ctx, const double *d_vals, double *S, double *d_rhs)
{
	SchurPlan &sp = ctx->schur;
	hipStream_t s = ctx->stream;
	const int64_t ld = sp.ld;
	double *xcol;
	if(sp.sparse_S) {
		// supernodal multifrontal factorization + solves of the sparse reduced camera system (the
		// reference's CLinearSolver_Schur with a sparse inner solver, LinearSolver_Schur.h:1844-1853)
		xcol = S + sp.s_st.nvals;
		const int ret = sparse_factor_solve(ctx, S, xcol);
		if(ret != SPP_OK)
			return ret;
	} else {
		phase_begin(ctx, SPP_PHASE_FACTOR);
		dense_set_padding(ctx, S, ld, sp.n_red);
		// the status of the factorization is fetched at the very end: a host round trip here would leave the GPU
		// idle for ~40 us before the solves (after a failure they run on garbage and the result is discarded)
		dense_potrf_upper_enqueue(ctx, S, sp.n_red, ld);
		phase_end(ctx, SPP_PHASE_FACTOR);
		xcol = S + sp.n_red * ld;
		phase_begin(ctx, SPP_PHASE_TRISOLVE);
		dense_potrs_upper(ctx, S, sp.n_red, ld, xcol);
		phase_end(ctx, SPP_PHASE_TRISOLVE);
	}
	phase_begin(ctx, SPP_PHASE_BACKSUBST);
	static_assert(DL <= DP, "the products U^T dx reuse the W l buffer (DP doubles per observation)");
	static int bs_fused = -1;
	if(bs_fused < 0) {
		const char *e = getenv("SPP_BACKSUBST_FUSED"); // 0: the products U^T dx through memory, two launches (rounds 1-3)
		bs_fused = e ? atoi(e) : 1;
	}
	const double *cinv_arg = (sp.factored && DL <= 3) ? (const double*)nullptr : (const double*)sp.cinv.p;
	if(bs_fused && sp.n_bs > 0)
		hipLaunchKernelGGL((backsubst_fused_kernel<DP, DL>), dim3((unsigned)sp.n_bs), dim3(256), 0, s,
			sp.bs_ptr.p, sp.lm_ptr.p, sp.obs_pose.p, sp.obs_off.p, sp.lm_rbase.p, cinv_arg, sp.lm_coff.p, d_vals, xcol, d_rhs);
	else {
		if(sp.no)
			hipLaunchKernelGGL((backsubst_obs_kernel<DP, DL>), dim3((unsigned)((sp.no + 255) / 256)), dim3(256), 0, s,
				sp.no, sp.obs_pose.p, sp.obs_off.p, d_vals, xcol, sp.xw.p);
		if(sp.nl)
			hipLaunchKernelGGL((backsubst_lm_kernel<DL>), dim3((unsigned)((sp.nl + 255) / 256)), dim3(256), 0, s,
				sp.nl, sp.lm_ptr.p, sp.lm_rbase.p, sp.xw.p, cinv_arg, sp.lm_coff.p, d_vals, d_rhs);
	}
	if(sp.nc)
		hipLaunchKernelGGL((scatter_dx_kernel<DP>), dim3((unsigned)((sp.nc * DP + 255) / 256)), dim3(256), 0, s,
			sp.nc, sp.pose_rbase.p, xcol, d_rhs);
	phase_end(ctx, SPP_PHASE_BACKSUBST);
	SPP_HIP_CHECK(hipGetLastError());
	if(!sp.sparse_S && dense_info_fetch(ctx))
		return SPP_NOT_POSDEF;
	return SPP_OK;
}

// packed upper trapezoid: panel k (columns 128 k ..) keeps rows [0, 128 (k + 1)), column-major inside
// the panel; panel offset = 128 * 128 * k (k + 1) / 2 doubles
template <bool PACK>
__global__ __launch_bounds__(256)
void schur_pack_kernel(double *__restrict__ S, int64_t ld, double *__restrict__ packed)
{
	const int64_t k = blockIdx.y;                 // panel
	const int64_t rows = 128 * (k + 1);
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; // element inside the panel (rows x 128)
	if(e >= rows * 128)
		return;
	const int64_t r = e % rows, c = e / rows;
	double *ps = S + r + (128 * k + c) * ld;
	double *pp = packed + 128 * 128 * (k * (k + 1) / 2) + e;
	if(PACK)
		*pp = *ps;
	else
		*ps = *pp;
}

void schur_pack(spp_ctx *ctx, double *S, double *packed, bool pack)
{
	const int64_t ld = ctx->schur.ld, nblk = ld / 128;
	dim3 grid((unsigned)((ld * 128 + 255) / 256), (unsigned)nblk);
	if(pack)
		hipLaunchKernelGGL((schur_pack_kernel<true>), grid, dim3(256), 0, ctx->stream, S, ld, packed);
	else
		hipLaunchKernelGGL((schur_pack_kernel<false>), grid, dim3(256), 0, ctx->stream, S, ld, packed);
	SPP_HIP_CHECK(hipGetLastError());
}

void schur_form(spp_ctx *ctx, const double *d_vals, const double *d_rhs, double *d_S_rhs)
{
	const int dp = ctx->schur.dp, dl = ctx->schur.dl;
	if(dp == 6 && dl == 3) schur_form_t<6, 3>(ctx, d_vals, d_rhs, d_S_rhs);
	else if(dp == 3 && dl == 2) schur_form_t<3, 2>(ctx, d_vals, d_rhs, d_S_rhs);
	else if(dp == 3 && dl == 3) schur_form_t<3, 3>(ctx, d_vals, d_rhs, d_S_rhs);
	else if(dp == 6 && dl == 6) schur_form_t<6, 6>(ctx, d_vals, d_rhs, d_S_rhs);
	else throw Error(SPP_E_UNSUPPORTED, "Schur kernels: block widths not instantiated");
}

int schur_finish(spp_ctx *ctx, const double *d_vals, double *d_S_rhs, double *d_rhs)
{
	const int dp = ctx->schur.dp, dl = ctx->schur.dl;
	if(dp == 6 && dl == 3) return schur_finish_t<6, 3>(ctx, d_vals, d_S_rhs, d_rhs);
	if(dp == 3 && dl == 2) return schur_finish_t<3, 2>(ctx, d_vals, d_S_rhs, d_rhs);
	if(dp == 3 && dl == 3) return schur_finish_t<3, 3>(ctx, d_vals, d_S_rhs, d_rhs);
	if(dp == 6 && dl == 6) return schur_finish_t<6, 6>(ctx, d_vals, d_S_rhs, d_rhs);
	throw Error(SPP_E_UNSUPPORTED, "Schur kernels: block widths not instantiated");
}

} // namespace spp
