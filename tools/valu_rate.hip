// issue rates / latencies of the f64 vector instructions on the pivot chain of the 16 x 16 diagonal tile (one wave):
// v_fma_f64 independent / dependent, v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, v_rcp_f64, v_readlane + SGPR-fed fma
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ long long g_t[16];
#define STAMP(t, val) do { union { double d; int i[2]; } w_; w_.d = (val); int s_ = __builtin_amdgcn_readfirstlane(w_.i[0]); \
	asm volatile("s_nop 0" :: "s"(s_)); __builtin_amdgcn_sched_barrier(0); t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while(0)
template <int L> __device__ __forceinline__ double rb(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false); }
__global__ __launch_bounds__(64) void k(double *out, double seed, int waves_busy)
{
	double a[16];
	for(int i = 0; i < 16; ++ i) a[i] = seed + threadIdx.x * 1e-9 + i;
	double y = 1.0000001, s = seed * 1e-9;
	long long t0, t1, t2, t3, t4, t5, t6, t7, t8;
	double sum = 0;
	for(int i = 0; i < 16; ++ i) sum += a[i];
	STAMP(t0, sum);
#pragma unroll
	for(int r = 0; r < 8; ++ r)
#pragma unroll
		for(int i = 0; i < 16; ++ i) a[i] = __builtin_fma(s, y, a[i]);  // 128 independent-ish (16 chains)
	sum = 0; for(int i = 0; i < 16; ++ i) sum += a[i];
	STAMP(t1, sum);
#pragma unroll
	for(int r = 0; r < 8; ++ r) {
#define F(I) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #I " row_mask:0xf bank_mask:0xf" : "+v"(a[I]) : "v"(s), "v"(y));
		F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
	}
	sum = 0; for(int i = 0; i < 16; ++ i) sum += a[i];
	STAMP(t2, sum);
	double x = sum;
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = __builtin_fma(x, y, 1e-9);
	STAMP(t3, x);
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = __builtin_amdgcn_rcp(x);
	STAMP(t4, x);
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = rb<3>(x) * y; // dpp mov + mul dependent
	STAMP(t5, x);
	// readlane pair + fma with an SGPR operand, 16 independent chains
#pragma unroll
	for(int r = 0; r < 8; ++ r)
#pragma unroll
		for(int i = 0; i < 16; ++ i) {
			union { double d; int w[2]; } u, v; u.d = s;
			v.w[0] = __builtin_amdgcn_readlane(u.w[0], i); v.w[1] = __builtin_amdgcn_readlane(u.w[1], i);
			a[i] = __builtin_fma(v.d, y, a[i]);
		}
	sum = 0; for(int i = 0; i < 16; ++ i) sum += a[i];
	STAMP(t6, sum);
	// two-instruction form: v_mov_b64_dpp + v_fma
#pragma unroll
	for(int r = 0; r < 8; ++ r) {
#define G(I) a[I] = __builtin_fma(rb<I>(s), y, a[I]);
		G(0) G(1) G(2) G(3) G(4) G(5) G(6) G(7) G(8) G(9) G(10) G(11) G(12) G(13) G(14) G(15)
	}
	sum = 0; for(int i = 0; i < 16; ++ i) sum += a[i];
	STAMP(t7, sum);
	// dependent mul chain
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = x * y;
	STAMP(t8, x);
	if(threadIdx.x == 0 && blockIdx.x == 0) { g_t[0] = t1 - t0; g_t[1] = t2 - t1; g_t[2] = t3 - t2; g_t[3] = t4 - t3; g_t[4] = t5 - t4; g_t[5] = t6 - t5; g_t[6] = t7 - t6; g_t[7] = t8 - t7; }
	out[threadIdx.x + 64 * blockIdx.x] = sum + x;
}
int main()
{
	double *d; (void)hipMalloc(&d, 64 * 8 * 16);
	for(int nw = 1; nw <= 4; nw *= 4) {
		for(int it = 0; it < 3; ++ it) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 0, 0, d, 1.5, nw); (void)hipDeviceSynchronize(); }
		long long t[16]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
		printf("%d wave(s) in the workgroup. cycles per op: fma_f64 indep %.1f  fmac_f64_dpp indep %.1f  fma_f64 dep %.1f  rcp_f64 dep %.1f  (mov_dpp + mul) dep %.1f  2 readlane + sgpr fma %.1f  mov_b64_dpp + fma %.1f  mul dep %.1f\n",
			nw, t[0] / 128.0, t[1] / 128.0, t[2] / 64.0, t[3] / 64.0, t[4] / 64.0, t[5] / 128.0, t[6] / 128.0, t[7] / 64.0);
	}
	return 0;
}
