// bisecting the per-pivot cost of the MFMA rank-1 loop (debug microbenchmark)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ long long g_t[16];
#define STAMP(t, val) do { union { double d; int i[2]; } w_; w_.d = (val); int s_ = __builtin_amdgcn_readfirstlane(w_.i[0]); \
	asm volatile("s_nop 0" :: "s"(s_)); __builtin_amdgcn_sched_barrier(0); t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while(0)
__global__ __launch_bounds__(64) void k(double *out, double seed, unsigned m0, unsigned m1)
{
	const int lane = threadIdx.x;
	double y = 1.0000001;
	v4f64 acc = {seed, seed + 1, seed + 2, seed + 3};
	long long t0, t1, t2, t3, t4, t5;
	double a = y, b = y;
	STAMP(t0, acc[0]);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // L2: sched barriers
		const double rv = acc[0];
		a = rv * 1e-30; b = rv * 1e-20;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t1, acc[0]);
	double keep = 0; int x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3; double pinv = 0;
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // M1: same component used by three multiplies
		const double rv = acc[0];
		a = rv * 1e-30; b = rv * 1e-20; double w = rv * 1e-10;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		keep += w;
	}
	STAMP(t2, acc[0] + keep);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // M2: two components, second one consumed in the shadow
		const double rv = acc[1], dg = acc[2];
		a = rv * 1e-30; b = rv * 1e-20;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		keep += dg * 1e-30;
	}
	STAMP(t3, acc[0] + keep);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // M3: + 4 independent integer VALU ops BEFORE the MFMA
		const double rv = acc[0];
		a = rv * 1e-30; b = rv * 1e-20;
		x0 &= m0 + i; x1 &= m1 + i; x2 &= m0 - i; x3 &= m1 - i;
		asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t4, acc[0] + x0 + x1 + x2 + x3);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // M4: + 4 independent integer VALU ops AFTER the MFMA (in its shadow)
		const double rv = acc[0];
		a = rv * 1e-30; b = rv * 1e-20;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		x0 &= m0 + i; x1 &= m1 + i; x2 &= m0 - i; x3 &= m1 - i;
		asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t5, acc[0] + x0 + x1 + x2 + x3);
	long long t6;
#pragma unroll
	for(int i = 0; i < 32; ++ i) { // M5: operands AND-masked with loop-invariant words
		const double rv = acc[0];
		union { double d; int i[2]; } ua, ub; ua.d = rv * 1e-30; ub.d = rv * 1e-20;
		ua.i[0] &= x0; ua.i[1] &= x0; ub.i[0] &= x1; ub.i[1] &= x1;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ua.d, ub.d, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t6, acc[0]);
	out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + keep + pinv;
	if(threadIdx.x == 0) { g_t[0] = t1 - t0; g_t[1] = t2 - t1; g_t[2] = t3 - t2; g_t[3] = t4 - t3; g_t[4] = t5 - t4; g_t[5] = t6 - t5; }
}
int main()
{
	double *d; (void)hipMalloc(&d, 64 * 8);
	for(int it = 0; it < 3; ++ it) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.5, 0x1111u, 0x2222u); (void)hipDeviceSynchronize(); }
	long long t[16]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
	printf("per MFMA: base %.1f  M1 three muls %.1f  M2 two comps %.1f  M3 4 int ops before %.1f  M4 4 int ops after %.1f  M5 and-masked %.1f\n",
		t[0] / 32.0, t[1] / 32.0, t[2] / 32.0, t[3] / 32.0, t[4] / 32.0, t[5] / 32.0);
	return 0;
}
