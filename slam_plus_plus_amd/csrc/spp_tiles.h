// spp_tiles.h -- 16 x 16 tile primitives on LDS images shared by the dense diagonal-block kernel
// (spp_dense.hip) and the in-LDS frontal kernels (spp_sparse.hip). gfx950, fp64 MFMA 16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// step A: factor + invert the 16 x 16 diagonal tile at (j0, j0) in registers (one wave).
// Lane (c = l & 15, g = l >> 4) owns rows 4g..4g+3 of column c. The strictly lower half accumulates
// G = (R_JJ^-1)^T:  pivot j, row i > j, f = W[j][i] / p_j:
//     c < j : W[i][c] -= f W[j][c]   (G update)      c == j: W[i][j] = -f   (new G entry)
//     c >= i: W[i][c] -= f W[j][c]   (trailing update)
// Writes R (upper) and G (strictly lower) into T, Dinv / G_JJ into the scratch tiles, 1/R_jj into dinv.
__device__ __forceinline__ void diag_tile_factor_rt(const int TS, double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
	const int l15 = lane & 15, l4 = lane >> 4;
	double x[4];
#pragma unroll
	for(int t = 0; t < 4; ++ t) {
		const int i = 4 * l4 + t;
		x[t] = (i <= l15) ? T[(j0 + i) + (j0 + l15) * TS] : 0.0;
	}
	bool bad = false;
#pragma unroll
	for(int j = 0; j < 16; ++ j) {
		const int src = j | ((j >> 2) << 4); // lane holding W[j][j] in register j & 3 (compile-time)
		const double p = readlane_f64(x[j & 3], src);
		if(!(p > 0)) {
			if(!bad && lane == 0) {
				*fail = 1;
				info[0] = (int)(k0 + j0 + j + 1);
			}
			bad = true;
		}
		double pinv = __builtin_amdgcn_rcp(p); // v_rcp_f64 + one Newton step
		pinv = pinv * (2.0 - p * pinv);
		const double rowj_c = __shfl(x[j & 3], l15 | ((j >> 2) << 4)); // W[j][c]
#pragma unroll
		for(int t = 0; t < 4; ++ t) {
			const int i = 4 * l4 + t;
			const double f = __shfl(x[j & 3], i | ((j >> 2) << 4)) * pinv; // W[j][i] / p
			const bool below = i > j;
			const double upd = x[t] - f * rowj_c;
			x[t] = (below && l15 == j) ? -f : ((below && (l15 < j || l15 >= i)) ? upd : x[t]);
		}
	}
	if(bad)
		return;
	double pv[4];
#pragma unroll
	for(int t = 0; t < 4; ++ t) {
		const int i = 4 * l4 + t;
		pv[t] = 1.0 / sqrt(__shfl(x[t], i | (l4 << 4))); // 1 / sqrt(W[i][i])
	}
#pragma unroll
	for(int t = 0; t < 4; ++ t) {
		const int i = 4 * l4 + t, c = l15;
		const double w = x[t], pi = pv[t];
		if(i < c) {          // R[i][c] = w / sqrt(p_i)
			T[(j0 + i) + (j0 + c) * TS] = w * pi;
			Dv[c + i * PT] = 0.0;   // Dinv[c][i], c > i: below the diagonal
			Gd[i + c * PT] = 0.0;   // G[i][c], c > i
		} else if(i == c) {
			T[(j0 + i) + (j0 + i) * TS] = 1.0 / pi;
			Dv[i + i * PT] = pi;
			Gd[i + i * PT] = pi;
			dinv[j0 + i] = pi;
		} else {             // G[i][c] = w / sqrt(p_i), c < i  (= Dinv[c][i])
			const double g = w * pi;
			T[(j0 + i) + (j0 + c) * TS] = g;
			Dv[c + i * PT] = g;
			Gd[i + c * PT] = g;
		}
	}
}


template <int TS>
__device__ __forceinline__ void tile_atb2(const double *a0, const double *b0, int b0ks, int b0js,
	const double *a1, const double *b1, int b1ks, int b1js, int lane, v4f64 &d0, v4f64 &d1)
{
	tile_atb2_rt(TS, a0, b0, b0ks, b0js, a1, b1, b1ks, b1js, lane, d0, d1);
}

template <int TS>
__device__ __forceinline__ void diag_tile_factor(double *T, double *Dv, double *Gd, double *dinv, int j0,
	int lane, int *fail, int *info, int64_t k0)
{
	diag_tile_factor_rt(TS, T, Dv, Gd, dinv, j0, lane, fail, info, k0);
}

} // namespace spp
