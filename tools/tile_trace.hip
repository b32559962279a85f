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
	for(int e = threadIdx.x; e < 16 * 17; e += 64) Tg[e] = T[e] + Dv[e] + Gd[e];
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
	(void)hipMalloc(&dA, 16 * 17 * 8); (void)hipMalloc(&dI, 16);
	for(int it = 0; it < 3; ++ it) {
		(void)hipMemcpy(dA, A.data(), 16 * 17 * 8, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dI);
		(void)hipDeviceSynchronize();
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
