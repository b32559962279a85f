// Issue rate of the fp64 MFMA forms with the operand pattern of a real tile product: a register outer product
// (distinct operand registers per instruction, random data), not one operand pair reused by every instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate2.hip -o tools/_bin/mfma_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// FORM 0: 16x16x4, 2 x 2 outer product (4 accumulators of 4 doubles); FORM 1: 4x4x4_4b, 8 x 2 outer product (16 accumulators)
template <int FORM>
__global__ __launch_bounds__(256)
void k(double *out, const double *in, int iters)
{
	double fn[8], fm[2];
	for(int i = 0; i < 8; ++ i) fn[i] = in[threadIdx.x + 256 * i];
	for(int i = 0; i < 2; ++ i) fm[i] = in[threadIdx.x + 256 * (8 + i)];
	v4f64 acc4[2][2];
	double acc1[8][2];
	for(int i = 0; i < 2; ++ i) for(int j = 0; j < 2; ++ j) acc4[i][j] = (v4f64){0, 0, 0, 0};
	for(int i = 0; i < 8; ++ i) for(int j = 0; j < 2; ++ j) acc1[i][j] = 0;
	for(int it = 0; it < iters; ++ it) {
		if(FORM == 0) {
#pragma unroll
			for(int r = 0; r < 4; ++ r) // 16 instructions = the flops of 64 of the small form
#pragma unroll
				for(int i = 0; i < 2; ++ i)
#pragma unroll
					for(int j = 0; j < 2; ++ j)
						acc4[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[i + 2 * r], fm[j], acc4[i][j], 0, 0, 0);
		} else {
#pragma unroll
			for(int r = 0; r < 4; ++ r)
#pragma unroll
				for(int i = 0; i < 8; ++ i)
#pragma unroll
					for(int j = 0; j < 2; ++ j)
						acc1[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fn[(i + r) & 7], fm[j], acc1[i][j], 0, 0, 0);
		}
		// keep the products bounded (and the operands changing): a cheap VALU op per iteration
		fm[0] = -fm[0];
	}
	double s = 0;
	for(int i = 0; i < 2; ++ i) for(int j = 0; j < 2; ++ j) s += acc4[i][j][0] + acc4[i][j][1] + acc4[i][j][2] + acc4[i][j][3];
	for(int i = 0; i < 8; ++ i) for(int j = 0; j < 2; ++ j) s += acc1[i][j];
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM>
void run(int waves_per_simd, double *out, const double *in)
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int ncu = p.multiProcessorCount, nblk = ncu * waves_per_simd, iters = 4000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL((k<FORM>), dim3(nblk), dim3(256), 0, 0, out, in, 16);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k<FORM>), dim3(nblk), dim3(256), 0, 0, out, in, iters);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double n_inst = (FORM == 0 ? 16.0 : 64.0) * iters * waves_per_simd; // per SIMD
	const double flop = (FORM == 0 ? 2048.0 : 512.0) * n_inst * 4 * ncu;
	printf("form %s  waves/SIMD %d : %.1f ns per MFMA and SIMD (%.0f cycles at 2.4 GHz), %.1f TFLOP/s (%.2f ms)\n",
		FORM == 0 ? "16x16x4  2x2 " : "4x4x4_4b 8x2 ", waves_per_simd, ms * 1e6 / n_inst, ms * 1e6 / n_inst * 2.4, flop / (ms * 1e-3) * 1e-12, ms);
}

int main()
{
	double *out, *in;
	hipMalloc(&out, sizeof(double) * 256 * 8 * 256 * 8);
	hipMalloc(&in, sizeof(double) * 256 * 10);
	double h[2560];
	unsigned s = 12345;
	for(int i = 0; i < 2560; ++ i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
	hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
	for(int rep = 0; rep < 2; ++ rep)
		for(int w = 1; w <= 8; w *= 2) {
			run<0>(w, out, in);
			run<1>(w, out, in);
		}
	return 0;
}
