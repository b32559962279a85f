// dependent-chain latencies of the instructions on the pivot chain (one wave): debug microbenchmark
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ long long g_t[16];
// the chain's last value is pulled into an SGPR before the clock is read: in-order issue then makes
// the stamp wait for the whole dependent chain
#define STAMP(t, val) do { union { double d; int i[2]; } w_; w_.d = (val); int s_ = __builtin_amdgcn_readfirstlane(w_.i[0]); \
	asm volatile("s_nop 0" :: "s"(s_)); __builtin_amdgcn_sched_barrier(0); t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while(0)
__global__ __launch_bounds__(64) void k(double *out, double seed)
{
	double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
	long long t0, t1, t2, t3, t4, t5, t6, t7;
	STAMP(t0, x);
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = __builtin_fma(x, y, 1e-9);
	STAMP(t1, x);
#pragma unroll
	for(int i = 0; i < 64; ++ i) x = __builtin_amdgcn_rcp(x);
	STAMP(t2, x);
	v4f64 acc = {x, x, x, x};
#pragma unroll
	for(int i = 0; i < 64; ++ i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
	STAMP(t3, acc[0]);
	double a = y;
#pragma unroll
	for(int i = 0; i < 64; ++ i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, y, acc, 0, 0, 0); a = acc[0] * 1e-30; }
	STAMP(t4, a);
	union { double d; int i[2]; } u; u.d = acc[1];
#pragma unroll
	for(int i = 0; i < 64; ++ i) { int s = __builtin_amdgcn_readlane(u.i[0], 5); u.i[0] = u.i[0] + s; }
	STAMP(t5, u.d);
	double z = u.d;
#pragma unroll
	for(int i = 0; i < 64; ++ i) z = z * y;
	STAMP(t6, z);
	float f = (float)z;
#pragma unroll
	for(int i = 0; i < 64; ++ i) f = __builtin_fmaf(f, 1.0001f, 1e-9f);
	STAMP(t7, (double)f);
	// (a) dependent MFMA chain + 8 independent f64 FMAs per MFMA; (b) + 8 independent v_readlane; (c) + dependent chain of 6 f64 ops
	long long t8, t9, t10, t11;
	double e0 = y, e1 = y + 1, e2 = y + 2, e3 = y + 3;
	STAMP(t8, acc[0]);
#pragma unroll
	for(int i = 0; i < 32; ++ i) {
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
		e0 = __builtin_fma(e0, y, 1e-9); e1 = __builtin_fma(e1, y, 1e-9); e2 = __builtin_fma(e2, y, 1e-9); e3 = __builtin_fma(e3, y, 1e-9);
		e0 = __builtin_fma(e0, y, 1e-9); e1 = __builtin_fma(e1, y, 1e-9); e2 = __builtin_fma(e2, y, 1e-9); e3 = __builtin_fma(e3, y, 1e-9);
	}
	STAMP(t9, acc[0] + e0 + e1 + e2 + e3);
	union { double d; int i[2]; } u2; u2.d = e0; int sacc = 0;
#pragma unroll
	for(int i = 0; i < 32; ++ i) {
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
		sacc += __builtin_amdgcn_readlane(u2.i[0], 1) + __builtin_amdgcn_readlane(u2.i[1], 2) + __builtin_amdgcn_readlane(u2.i[0], 3) + __builtin_amdgcn_readlane(u2.i[1], 4);
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t10, acc[0] + sacc);
#pragma unroll
	for(int i = 0; i < 32; ++ i) {
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
		e0 = __builtin_fma(e0, y, 1e-9); e0 = __builtin_fma(e0, y, 1e-9); e0 = __builtin_amdgcn_rcp(e0); e0 = __builtin_fma(e0, y, 1e-9); e0 = __builtin_fma(e0, y, 1e-9); e0 = __builtin_fma(e0, y, 1e-9);
		__builtin_amdgcn_sched_barrier(0);
	}
	STAMP(t11, acc[0] + e0);
	long long t12, t13, t14, t15, t16;
	double b = y;
	a = y;
#pragma unroll
	for(int i = 0; i < 32; ++ i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0] * 1e-30; b = acc[0] * 1e-20; }
	STAMP(t12, a + b);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, y, acc, 0, 0, 0); a = acc[(i & 3)] * 1e-30; }
	STAMP(t13, a);
#pragma unroll
	for(int i = 0; i < 32; ++ i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		a = (threadIdx.x > (unsigned)i) ? acc[0] * 1e-30 : 0.0; b = (threadIdx.x == (unsigned)i) ? 0.0 : acc[0]; }
	STAMP(t14, a + b);
	sacc = 0;
#pragma unroll
	for(int i = 0; i < 32; ++ i) { 
		const double rv = acc[0];
		a = rv * 1e-30; b = rv;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		union { double d; int i[2]; } w; w.d = rv;
		sacc += __builtin_amdgcn_readlane(w.i[0], 1) + __builtin_amdgcn_readlane(w.i[1], 2); }
	STAMP(t15, a + b + sacc);
	sacc = 0;
#pragma unroll
	for(int i = 0; i < 32; ++ i) {
		const double rv = acc[0];
		a = rv * 1e-30; b = rv;
		union { double d; int i[2]; } w; w.d = rv;
		const int s0 = __builtin_amdgcn_readlane(w.i[0], 1), s1 = __builtin_amdgcn_readlane(w.i[1], 2);
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		sacc += s0 + s1; }
	STAMP(t16, a + b + sacc);
	// readlane results consumed by VALU (VGPR chain) in the shadow
	double zz = 1.0;
#pragma unroll
	for(int i = 0; i < 32; ++ i) {
		const double rv = acc[0];
		a = rv * 1e-30 * zz; b = rv;
		__builtin_amdgcn_sched_barrier(0);
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		union { double d; int i[2]; } w, w2; w.d = rv;
		w2.i[0] = __builtin_amdgcn_readlane(w.i[0], 1); w2.i[1] = __builtin_amdgcn_readlane(w.i[1], 1);
		zz = __builtin_fma(w2.d, 1e-30, zz); zz = __builtin_amdgcn_rcp(zz); zz = zz * (2.0 - zz); }
	long long t17; STAMP(t17, a + b + zz);
	if(threadIdx.x == 0) { g_t[15] = t16 - t15; g_t[7] = t17 - t16; }
	if(threadIdx.x == 0) { g_t[11] = t12 - t11; g_t[12] = t13 - t12; g_t[13] = t14 - t13; g_t[14] = t15 - t14; }
	if(threadIdx.x == 0) { g_t[8] = t9 - t8; g_t[9] = t10 - t9; g_t[10] = t11 - t10; }
	out[threadIdx.x] = e0 + e1 + e2 + e3 + sacc + acc[0] + acc[1] + a + u.d + z + f;
	if(threadIdx.x == 0) { g_t[0] = t1 - t0; g_t[1] = t2 - t1; g_t[2] = t3 - t2; g_t[3] = t4 - t3; g_t[4] = t5 - t4; g_t[5] = t6 - t5; g_t[6] = t7 - t6; }
}
int main()
{
	double *d; (void)hipMalloc(&d, 64 * 8);
	for(int it = 0; it < 3; ++ it) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.5); (void)hipDeviceSynchronize(); }
	long long t[16]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof(t));
	printf("per op (cycles): fma_f64 %.1f  rcp_f64 %.1f  mfma_f64 dep-acc %.1f  mfma->mul->mfma %.1f  readlane+add %.1f  mul_f64 %.1f  fma_f32 %.1f\n",
		t[0] / 64.0, t[1] / 64.0, t[2] / 64.0, t[3] / 64.0, t[4] / 64.0, t[5] / 64.0, t[6] / 64.0);
	printf("per MFMA (cycles): + 8 indep fma_f64 %.1f   + 4 readlane %.1f   + dependent 6-op f64 chain with rcp %.1f\n", t[8] / 32.0, t[9] / 32.0, t[10] / 32.0);
	printf("per MFMA: both operands from acc %.1f   rotating acc component %.1f   masked operands %.1f   + readlanes in shadow %.1f\n", t[11] / 32.0, t[12] / 32.0, t[13] / 32.0, t[14] / 32.0);
	printf("per MFMA: readlanes BEFORE the mfma, SALU use after %.1f   readlane in shadow feeding a VALU chain %.1f\n", t[15] / 32.0, t[7] / 32.0);
	return 0;
}
