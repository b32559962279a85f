// cycle stamps inside diag_tile_factor (debug harness): one wave, one 16 x 16 SPD tile in LDS
#include <hip/hip_runtime.h>
__device__ long long g_stamp[16], g_piv[16];
#ifdef PIVSTAMP
#define SPP_PIVOT_STAMP(j, val) do { union { double d; int i[2]; } w_; w_.d = (val); int s_ = __builtin_amdgcn_readfirstlane(w_.i[0]); \
	asm volatile("s_nop 0" :: "s"(s_)); __builtin_amdgcn_sched_barrier(0); long long t_ = __builtin_readcyclecounter(); \
	__builtin_amdgcn_sched_barrier(0); if(lane == 0) g_piv[j] = t_; } while(0)
#endif
#define SPP_TILE_STAMP(k, val) do { union { double d; int i[2]; } w_; w_.d = (val); int s_ = __builtin_amdgcn_readfirstlane(w_.i[0] ^ w_.i[1]); \
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" :: "s"(s_) : "memory"); __builtin_amdgcn_sched_barrier(0); long long t_ = __builtin_readcyclecounter(); \
	__builtin_amdgcn_sched_barrier(0); if(lane == 0) g_stamp[k] = t_; } while(0)
#include "../slam_plus_plus_amd/csrc/spp_tiles.h"
#include <stdio.h>
#include <math.h>
#include <vector>
#include <random>

__global__ __launch_bounds__(64) void k(double *Tg, int *info)
{
	__shared__ double T[16 * 17], Dv[16 * 17], Gd[16 * 17], dinv[16];
	__shared__ int fail;
	for(int e = threadIdx.x; e < 16 * 17; e += 64) T[e] = Tg[e];
	__syncthreads();
	spp::diag_tile_factor<17>(T, Dv, Gd, dinv, 0, threadIdx.x, &fail, info, 0);
	__syncthreads();
	for(int e = threadIdx.x; e < 16 * 17; e += 64) { Tg[e] = T[e]; Tg[16 * 17 + e] = Dv[e]; Tg[2 * 16 * 17 + e] = Gd[e]; }
}

int main()
{
	const int n = 16;
	std::vector<double> A(16 * 17, 0.0), M(n * n);
	std::mt19937_64 g(1);
	std::normal_distribution<double> nd;
	for(auto &v : M) v = nd(g);
	for(int i = 0; i < n; ++ i)
		for(int j = 0; j < n; ++ j) {
			double s = 0;
			for(int kk = 0; kk < n; ++ kk) s += M[i + kk * n] * M[j + kk * n];
			A[i + j * 17] = s / n + (i == j ? 2.0 : 0.0);
		}
	double *dA; int *dI;
	(void)hipMalloc(&dA, 3 * 16 * 17 * 8); (void)hipMalloc(&dI, 16);
	for(int it = 0; it < 3; ++ it) {
		(void)hipMemcpy(dA, A.data(), 16 * 17 * 8, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dI);
		(void)hipDeviceSynchronize();
	}
	{
		// check against a host factorization: T upper = R, T strictly lower = G = (R^-1)^T, Dv[c + i*17] = Gd[i + c*17] = Rinv[c][i] (i >= c)
		std::vector<double> out(3 * 16 * 17), R(n * n, 0.0), Ri(n * n, 0.0);
		(void)hipMemcpy(out.data(), dA, 3 * 16 * 17 * 8, hipMemcpyDeviceToHost);
		for(int j = 0; j < n; ++ j) { // R^T R = A, R upper, R(i, j) at R[i + j * n]
			for(int i = 0; i <= j; ++ i) {
				double s = A[i + j * 17];
				for(int kk = 0; kk < i; ++ kk) s -= R[kk + i * n] * R[kk + j * n];
				R[i + j * n] = (i == j) ? sqrt(s) : s / R[i + i * n];
			}
		}
		for(int j = 0; j < n; ++ j) { // Ri = R^-1 (upper): back substitution per unit vector
			for(int i = j; i >= 0; -- i) {
				double s = (i == j) ? 1.0 : 0.0;
				for(int kk = i + 1; kk <= j; ++ kk) s -= R[i + kk * n] * Ri[kk + j * n];
				Ri[i + j * n] = s / R[i + i * n];
			}
		}
		double eR = 0, eG = 0, eD = 0, eGd = 0;
		for(int cc = 0; cc < n; ++ cc)
			for(int i = 0; i < n; ++ i) {
				const double t = out[i + cc * 17];
				if(i <= cc) eR = fmax(eR, fabs(t - R[i + cc * n]));
				else eG = fmax(eG, fabs(t - Ri[cc + i * n])); // G[i][c] = Rinv[c][i]
				const double want = (i >= cc) ? Ri[cc + i * n] : 0.0;
				eD = fmax(eD, fabs(out[16 * 17 + cc + i * 17] - want));
				eGd = fmax(eGd, fabs(out[2 * 16 * 17 + i + cc * 17] - want));
			}
		printf("max abs err: R %.2e  G %.2e  Dinv %.2e  Gd %.2e\n", eR, eG, eD, eGd);
	}
	long long t[16];
	(void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stamp), sizeof(t));
	printf("load %lld  pivots %lld  sqrt/exchange %lld  store %lld  total %lld\n", t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0]);
	long long pv[16];
	(void)hipMemcpyFromSymbol(pv, HIP_SYMBOL(g_piv), sizeof(pv));
	printf("per pivot:");
	for(int j = 1; j < 15; ++ j) printf(" %lld", pv[j] - pv[j - 1]);
	printf("\n");
	return 0;
}
