// spp_tiles.h -- 16 x 16 tile primitives on LDS images shared by the dense diagonal-block kernel
// (spp_dense.hip) and the in-LDS frontal kernels (spp_sparse.hip). gfx950, fp64 MFMA 16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef SPP_TILE_STAMP
#define SPP_TILE_STAMP(k, v) do { } while(0)
#endif
#ifndef SPP_PIVOT_STAMP
#define SPP_PIVOT_STAMP(j, v) do { } while(0)
#endif

namespace spp {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int PT = 17;       // column stride of the 16 x 16 scratch tiles

// wave-uniform broadcast of a double from a compile-time lane (two v_readlane_b32)
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
	union { double d; int i[2]; } u;
	u.d = v;
	u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
	u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
	return u.d;
}

// broadcast of a double from a compile-time lane to ALL lanes, result in VGPRs (two ds_bpermute_b32 through the
// LDS crossbar, no memory access): asynchronous (lgkmcnt), so it can be issued ahead of an MFMA and waited for in
// its shadow -- unlike v_readlane, which writes SGPRs and stalls 70-100 cycles while the wave's MFMA is in flight
__device__ __forceinline__ double bcast_f64(double v, int src_lane)
{
	union { double d; int i[2]; } u;
	u.d = v;
	u.i[0] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[0]);
	u.i[1] = __builtin_amdgcn_ds_bpermute(src_lane << 2, u.i[1]);
	return u.d;
}

// bit j of a per-lane mask as 0 / ~0 (v_bfe_i32), and a double AND-ed with such a word: VGPR-only masking
__device__ __forceinline__ int sbit(unsigned mask, int j)
{
	return ((int)(mask << (31 - j))) >> 31;
}

__device__ __forceinline__ double and_f64(double v, int word)
{
	union { double d; int i[2]; } u;
	u.d = v;
	u.i[0] &= word;
	u.i[1] &= word;
	return u.d;
}

// D (16 x 16) = sum_k A[k][i] * B[k][j], k = 0..15; A element (k, i) at a[k * aks + i * ais],
// B element (k, j) at b[k * bks + j * bjs]. Result in the MFMA D layout (row (l>>4) + 4 r, col l & 15).

__device__ __forceinline__ v4f64 tile_atb(const double *a, int aks, int ais, const double *b, int bks, int bjs, int lane)
{
	v4f64 acc = (v4f64){0, 0, 0, 0};
	const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
	for(int kk = 0; kk < 4; ++ kk) {
		const double fa = a[(kk * 4 + l4) * aks + l15 * ais];
		const double fb = b[(kk * 4 + l4) * bks + l15 * bjs];
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc, 0, 0, 0);
	}
	return acc;
}

// two independent tiles at once: the loads and MFMA chains of both interleave
__device__ __forceinline__ void tile_atb2_rt(const int TS, const double *a0, const double *b0, int b0ks, int b0js,
	const double *a1, const double *b1, int b1ks, int b1js, int lane, v4f64 &d0, v4f64 &d1)
{
	d0 = (v4f64){0, 0, 0, 0};
	d1 = (v4f64){0, 0, 0, 0};
	const int l15 = lane & 15, l4 = lane >> 4;
	double fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
	for(int kk = 0; kk < 4; ++ kk) {
		fa0[kk] = a0[(kk * 4 + l4) + l15 * TS];
		fb0[kk] = b0[(kk * 4 + l4) * b0ks + l15 * b0js];
		fa1[kk] = a1[(kk * 4 + l4) + l15 * TS];
		fb1[kk] = b1[(kk * 4 + l4) * b1ks + l15 * b1js];
	}
#pragma unroll
	for(int kk = 0; kk < 4; ++ kk) {
		d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[kk], fb0[kk], d0, 0, 0, 0);
		d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[kk], fb1[kk], d1, 0, 0, 0);
	}
}

// three independent tiles at once (operand element (k, i) of a tile at a[k + i * TS], (k, j) at b[k + j * bjs])
__device__ __forceinline__ void tile_atb3_rt(const int TS, const double *a0, const double *b0, int b0js,
	const double *a1, const double *b1, int b1js, const double *a2, const double *b2, int b2js, int lane,
	v4f64 &d0, v4f64 &d1, v4f64 &d2)
{
	d0 = (v4f64){0, 0, 0, 0};
	d1 = (v4f64){0, 0, 0, 0};
	d2 = (v4f64){0, 0, 0, 0};
	const int l15 = lane & 15, l4 = lane >> 4;
	double fa0[4], fb0[4], fa1[4], fb1[4], fa2[4], fb2[4];
#pragma unroll
	for(int kk = 0; kk < 4; ++ kk) {
		fa0[kk] = a0[(kk * 4 + l4) + l15 * TS];
		fb0[kk] = b0[(kk * 4 + l4) + l15 * b0js];
		fa1[kk] = a1[(kk * 4 + l4) + l15 * TS];
		fb1[kk] = b1[(kk * 4 + l4) + l15 * b1js];
		fa2[kk] = a2[(kk * 4 + l4) + l15 * TS];
		fb2[kk] = b2[(kk * 4 + l4) + l15 * b2js];
	}
#pragma unroll
	for(int kk = 0; kk < 4; ++ kk) {
		d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[kk], fb0[kk], d0, 0, 0, 0);
		d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[kk], fb1[kk], d1, 0, 0, 0);
		d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa2[kk], fb2[kk], d2, 0, 0, 0);
	}
}

// step A: factor + invert the 16 x 16 diagonal tile at (j0, j0) in registers (one wave).
// The symmetric tile W lives in the MFMA accumulator layout: lane (c = l & 15, q = l >> 4) holds
// W[q + 4 r][c] in acc[r]. The rank-1 update of pivot j,
//     W[i][c] -= (W[j][i] / p_j) W[j][c]      rows i > j, columns c != j,
// is ONE v_mfma_f64_16x16x4: row j sits in the lanes q == (j & 3) (register j >> 2), which feed it
// unmoved as both operands (A[i][k] = -W[j][i] / p_j for i > j, B[k][c] = W[j][c] for c != j, zero in
// the other three k slots) -- the matrix core does the broadcast, no lane shuffles, no LDS.
// The next pivot W[j+1][j+1] - (W[j][j+1] / p_j) W[j][j+1] is formed beside the MFMA from two
// v_readlane'd scalars, so its reciprocal (v_rcp_f64 + one Newton step) overlaps the MFMA; the serial
// chain per pivot is one MFMA plus the operand scaling.
// Rows i > c of column c first carry the mirrored (lower) half of the trailing matrix; column c is left
// out of its own pivot, and from then on those rows evolve, through the very same update, into column
// c of G = (R_JJ^-1)^T up to the factor -1 / p_c, applied once at the end (the updates are linear in
// the column, so the scaling commutes with them).
// Writes R (upper) and G (strictly lower) into T, Dinv / G_JJ into the scratch tiles, 1/R_jj into dinv.
__device__ __forceinline__ void diag_tile_factor_rt(const int TS, double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
	const int c = lane & 15, q = lane >> 4;
	SPP_TILE_STAMP(0, 0.0);
	v4f64 acc;
#pragma unroll
	for(int r = 0; r < 4; ++ r) {
		const int i = q + 4 * r;
		acc[r] = T[(i <= c) ? (j0 + i) + (j0 + c) * TS : (j0 + c) + (j0 + i) * TS];
	}
	// Measured on gfx950 (tools/lat_bench.hip, lat_bench2.hip): a dependent v_mfma_f64_16x16x4 chain with
	// both operands rebuilt from the accumulator runs at ~105 cycles per MFMA; a VALU instruction that
	// writes an SGPR while the wave's MFMA is in flight (v_readlane, v_cmp) adds 70-100 cycles, a
	// compare + select mask pair ~15. Hence: the two v_readlane'd scalars of the next pivot are taken
	// BEFORE the MFMA is issued, the lane masks are per-lane bit sets built before the loop and applied
	// as AND words, and the pivot test runs once after the loop on the diagonal itself.
	// (bit-AND with 0 / ~0: exact zeros even when a failed pivot has produced NaNs; after an exactly zero
	// pivot the NaNs still spread through B, the factorization is then reported as failed, at worst
	// with an earlier pivot index than the true one).
	//   A mask, bit j: lane feeds row i = c > j in k slot q == (j & 3);  B mask, bit j: not (slot q, column j)
	const unsigned amask = (0x1111u << q) & ((1u << c) - 1u);
	const unsigned bmask = ~(((c & 3) == q) ? (1u << c) : 0u);
	double p = readlane_f64(acc[0], 0);
	SPP_TILE_STAMP(1, p);
	double r0 = __builtin_amdgcn_rcp(p);
	double pinv = r0 * (2.0 - p * r0); // v_rcp_f64 + one Newton step
	int aw = sbit(amask, 0), bw = sbit(bmask, 0);
#pragma unroll
	for(int j = 0; j < 15; ++ j) {
		const int qj = j & 3, rj = j >> 2;
		// ---- between two MFMAs (the serial chain): as few instructions as possible
		const double rowv = acc[rj];
		const double diag = acc[(j + 1) >> 2];
		const double aop = and_f64(-rowv * pinv, aw); // A[i][k]: -W[j][i] / p_j for rows i > j
		const double bop = and_f64(rowv, bw);         // B[k][c]:  W[j][c], column j dropped
		// scalars of the next pivot: W[j][j+1] and W[j+1][j+1] before this pivot's update
#ifndef SPP_TILE_BPERMUTE
		const double wj = readlane_f64(rowv, (j + 1) | (qj << 4));
		const double wd = readlane_f64(diag, (j + 1) | (((j + 1) & 3) << 4));
#else
		// experiment: broadcast into VGPRs (ds_bpermute), issued before the MFMA and consumed in its shadow. Measured
		// 5 700 instead of 5 400 cycles per tile: the crossbar round trip plus fma -> rcp -> Newton -> scale is longer
		// than the MFMA it was meant to hide under. Kept for reference.
		const double wj = bcast_f64(rowv, (j + 1) | (qj << 4));
		const double wd = bcast_f64(diag, (j + 1) | (((j + 1) & 3) << 4));
#endif
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		// ---- in the MFMA's shadow (VGPR results only): the next pivot, rounded exactly like the MFMA's
		// own update of W[j+1][j+1], its reciprocal, and the next mask words
		p = __builtin_fma(-wj * pinv, wj, wd);
		r0 = __builtin_amdgcn_rcp(p);
		pinv = r0 * (2.0 - p * r0);
		aw = sbit(amask, j + 1);
		bw = sbit(bmask, j + 1);
		asm volatile("" : "+v"(pinv), "+v"(aw), "+v"(bw)); // keep these before the wait for the MFMA
		__builtin_amdgcn_sched_barrier(0);
		SPP_PIVOT_STAMP(j, p);
	}
	SPP_TILE_STAMP(2, acc[0] + acc[1] + acc[2] + acc[3] + p);
	// the lanes with (c & 3) == q hold the diagonal p_c in acc[c >> 2]; 1 / sqrt(p_i) is exchanged via dinv
	double pc = acc[0];
#pragma unroll
	for(int r = 1; r < 4; ++ r)
		pc = ((c >> 2) == r) ? acc[r] : pc;
	// Eigen's LLT test (non-positive or NaN pivot) on the diagonal lanes; the first failing column is
	// exact because the pivots before it do not depend on it
	{
		const unsigned long long bl = __builtin_amdgcn_ballot_w64(((c & 3) == q) && !(pc > 0));
		if(bl) {
			const unsigned m16 = (unsigned)((bl & 0x1111u) | ((bl >> 16) & 0x2222u) | ((bl >> 32) & 0x4444u) | ((bl >> 48) & 0x8888u));
			if(lane == 0) {
				*fail = 1;
				info[0] = (int)(k0 + j0 + __builtin_ctz(m16) + 1);
			}
			return;
		}
	}
	const double sq = sqrt(pc); // R_cc (meaningful in the diagonal lanes)
	if((c & 3) == q)
		dinv[j0 + c] = 1.0 / sq;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const double pvc = dinv[j0 + c];
	SPP_TILE_STAMP(3, pvc);
	const double g = -pvc * pvc; // -1 / p_c: the deferred scale of column c of G
#pragma unroll
	for(int r = 0; r < 4; ++ r) {
		const int i = q + 4 * r;
		const double pvi = dinv[j0 + i];
		// i < c: R[i][c] = w / sqrt(p_i);  i > c: G[i][c] = (-w / p_c) / sqrt(p_i) = Dinv[c][i]
		const double w = (i == c) ? pvc : acc[r] * pvi * ((i > c) ? g : 1.0);
		T[(j0 + i) + (j0 + c) * TS] = (i == c) ? sq : w;
		const double d = (i < c) ? 0.0 : w;
		Dv[c + i * PT] = d;   // Dinv[c][i] (upper triangular, zero for c > i)
		Gd[i + c * PT] = d;   // G[i][c]   (lower triangular incl. the diagonal)
	}
	SPP_TILE_STAMP(4, 0.0);
}


// --------------------------------------------------------------------------------------------------
// The same elimination on the vector ALU with DPP row broadcasts (gfx90a+: row_newbcast, the one DPP
// control that 64-bit operations accept). Lane c of every 16-lane row holds the WHOLE column c of the
// symmetric tile, A[i] = W[i][c] (32 VGPRs; the four rows of 16 lanes compute redundantly, which costs
// nothing: an instruction takes the same cycles for 16 active lanes as for 64). The rank-1 update of pivot j,
//     W[i][c] -= (W[j][i] / p_j) W[j][c]      rows i > j, columns c != j,
// is ONE v_fmac_f64_dpp per row i: the multiplier -W[j][i] / p_j is lane i's entry of the scaled pivot row
// s = -A[j] / p_j, fetched by row_newbcast:i, the other factor is the lane's own A[j] (AND-masked to zero in
// lane j, which keeps column j out of its own pivot exactly as in the MFMA formulation above, so that rows
// i > c of column c evolve into column c of G). The rows are independent of each other: 15 - j instructions
// of 4..8 cycles instead of one dependent v_mfma_f64_16x16x4 (~105 cycles) plus ~140 cycles of operand set-up,
// and no SGPR-writing v_readlane in the shadow of an MFMA. The next pivot is formed ahead of the update from
// two broadcast scalars with the very fma the update applies to W[j+1][j+1], so its reciprocal
// (v_rcp_f64 + one Newton step) runs beside the row updates. Same outputs, same algebra, different rounding
// order inside a row only through fma contraction (none: every update is one fma in both forms).
// Hazard: a DPP source written by the preceding VALU instruction needs two wait states; the compiler pads
// nothing inside an asm statement, so the first update of a pivot carries its own s_nop 1.
template <int L>
__device__ __forceinline__ double row_bcast_f64(double v)
{
	return __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false); // v_mov_b64_dpp row_newbcast:L (every lane is written: no separate `old`)
}

template <int I, int NOP>
__device__ __forceinline__ void fmac_row_bcast(double &a, const double s, const double b)
{
	if(NOP)
		asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
			: "+v"(a) : "v"(s), "v"(b), "n"(I));
	else
		asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
			: "+v"(a) : "v"(s), "v"(b), "n"(I));
}

template <int J, int I>
struct DiagRowUpdate {
	static __device__ __forceinline__ void run(double (&A)[16], const double s, const double b)
	{
		fmac_row_bcast<I, (I == J + 1)>(A[I], s, b);
		DiagRowUpdate<J, I + 1>::run(A, s, b);
	}
};
template <int J>
struct DiagRowUpdate<J, 16> {
	static __device__ __forceinline__ void run(double (&)[16], const double, const double) {}
};

template <int J>
struct DiagPivot {
	static __device__ __forceinline__ void run(double (&A)[16], double &pinv, double &pc, const unsigned nmask, const int c)
	{
		[[maybe_unused]] const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); // (stamps only)
		const double rowj = A[J];
		const double s = -rowj * pinv;                     // lane c: -W[j][c] / p_j
		const double bj = and_f64(rowj, sbit(nmask, J));   // W[j][c], zero in lane j
		// scalars of the next pivot, before this pivot's update: W[j][j+1] and W[j+1][j+1] live in lane j+1
		const double wj = row_bcast_f64<J + 1>(rowj);
		const double wd = row_bcast_f64<J + 1>(A[J + 1]);
		DiagRowUpdate<J, J + 1>::run(A, s, bj);
		// rounded exactly like the update of W[j+1][j+1] itself: fma(-(W[j][j+1] p_j^-1), W[j][j+1], W[j+1][j+1])
		const double p = __builtin_fma(-wj * pinv, wj, wd);
		const double r0 = __builtin_amdgcn_rcp(p);
		pinv = r0 * (2.0 - p * r0);
		pc = (c == J + 1) ? p : pc; // lane j+1 keeps its pivot
		SPP_PIVOT_STAMP(J, p);
		DiagPivot<J + 1>::run(A, pinv, pc, nmask, c);
	}
};
template <>
struct DiagPivot<15> {
	static __device__ __forceinline__ void run(double (&)[16], double &, double &, const unsigned, const int) {}
};

__device__ __forceinline__ void diag_tile_factor_dpp_rt(const int TS, double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
	const int c = lane & 15, q = lane >> 4;
	SPP_TILE_STAMP(0, 0.0);
	double A[16];
#pragma unroll
	for(int i = 0; i < 16; ++ i)
		A[i] = T[(i <= c) ? (j0 + i) + (j0 + c) * TS : (j0 + c) + (j0 + i) * TS];
	const unsigned nmask = ~(1u << c);
	double pc = row_bcast_f64<0>(A[0]); // p_0 in every lane; lane c ends up with p_c
	SPP_TILE_STAMP(1, pc);
	double r0 = __builtin_amdgcn_rcp(pc);
	double pinv = r0 * (2.0 - pc * r0); // v_rcp_f64 + one Newton step
	DiagPivot<0>::run(A, pinv, pc, nmask, c);
	SPP_TILE_STAMP(2, A[15] + pc);
	// Eigen's LLT test (non-positive or NaN pivot) on the pivots themselves (the lanes of the first row of 16);
	// the first failing column is exact because the pivots before it do not depend on it
	{
		const unsigned long long bl = __builtin_amdgcn_ballot_w64(q == 0 && !(pc > 0));
		if(bl) {
			if(lane == 0) {
				*fail = 1;
				info[0] = (int)(k0 + j0 + __builtin_ctzll(bl) + 1);
			}
			return;
		}
	}
	const double sq = sqrt(pc); // R_cc
	if(q == 0)
		dinv[j0 + c] = 1.0 / sq;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const double pvc = dinv[j0 + c];
	SPP_TILE_STAMP(3, pvc);
	const double g = -pvc * pvc; // -1 / p_c: the deferred scale of column c of G
	// the four rows of 16 lanes share the write-back: lane (q, c) finishes rows i = q + 4 r
#pragma unroll
	for(int r = 0; r < 4; ++ r) {
		const int i = q + 4 * r;
		// (the values are pinned in registers first: a select between two array elements is otherwise turned into an
		// indexed load and the whole column goes through scratch memory)
		double x0 = A[4 * r], x1 = A[4 * r + 1], x2 = A[4 * r + 2], x3 = A[4 * r + 3];
		asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
		const double a01 = (q & 1) ? x1 : x0, a23 = (q & 1) ? x3 : x2;
		const double a = (q & 2) ? a23 : a01;
		const double pvi = dinv[j0 + i];
		// i < c: R[i][c] = w / sqrt(p_i);  i > c: G[i][c] = (-w / p_c) / sqrt(p_i) = Dinv[c][i]
		const double w = (i == c) ? pvc : a * pvi * ((i > c) ? g : 1.0);
		T[(j0 + i) + (j0 + c) * TS] = (i == c) ? sq : w;
		const double d = (i < c) ? 0.0 : w;
		Dv[c + i * PT] = d;   // Dinv[c][i] (upper triangular, zero for c > i)
		Gd[i + c * PT] = d;   // G[i][c]   (lower triangular incl. the diagonal)
	}
	SPP_TILE_STAMP(4, 0.0);
}

#ifndef SPP_DIAG_DPP
#define SPP_DIAG_DPP 1 // 0: the MFMA formulation (diag_tile_factor_rt above), kept for A/B timing
#endif

template <int TS>
__device__ __forceinline__ void tile_atb2(const double *a0, const double *b0, int b0ks, int b0js,
	const double *a1, const double *b1, int b1ks, int b1js, int lane, v4f64 &d0, v4f64 &d1)
{
	tile_atb2_rt(TS, a0, b0, b0ks, b0js, a1, b1, b1ks, b1js, lane, d0, d1);
}

// the form in use (TS: run-time or compile-time LDS stride of the image)
__device__ __forceinline__ void diag_tile_factor_sel(const int TS, double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
#if SPP_DIAG_DPP
	diag_tile_factor_dpp_rt(TS, T, Dv, Gd, dinv, j0, lane, fail, info, k0);
#else
	diag_tile_factor_rt(TS, T, Dv, Gd, dinv, j0, lane, fail, info, k0);
#endif
}

template <int TS>
__device__ __forceinline__ void diag_tile_factor(double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
	diag_tile_factor_sel(TS, T, Dv, Gd, dinv, j0, lane, fail, info, k0);
}

} // namespace spp
