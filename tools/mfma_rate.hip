// Issue rate of the fp64 MFMA forms on gfx950: cycles per instruction and SIMD as a function of waves per SIMD and of the
// number of independent accumulators per wave (hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/_bin/mfma_rate).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC, int FORM>
__global__ __launch_bounds__(256)
void k(double *out, int iters)
{
	v4f64 acc[NACC];
	double acc1[NACC];
#pragma unroll
	for(int i = 0; i < NACC; ++ i) {
		acc[i] = (v4f64){0, 0, 0, 0};
		acc1[i] = 0;
	}
	double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
	for(int it = 0; it < iters; ++ it) {
#pragma unroll
		for(int i = 0; i < NACC; ++ i) {
			if(FORM == 0)
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
			else
				acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
		}
	}
	double s = 0;
#pragma unroll
	for(int i = 0; i < NACC; ++ i)
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FORM>
void run(int waves_per_simd, double *out)
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int ncu = p.multiProcessorCount, nblk = ncu * waves_per_simd, iters = 20000 / NACC;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL((k<NACC, FORM>), dim3(nblk), dim3(256), 0, 0, out, 16);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k<NACC, FORM>), dim3(nblk), dim3(256), 0, 0, out, iters);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double n_per_simd = (double)waves_per_simd * iters * NACC; // MFMAs each SIMD issues
	const double flop = (FORM == 0 ? 2048.0 : 512.0) * n_per_simd * 4 * ncu;
	printf("form %s  waves/SIMD %d  accumulators %2d : %.1f ns per MFMA and SIMD (%.0f cycles at 2.4 GHz), %.1f TFLOP/s\n",
		FORM == 0 ? "16x16x4   " : "4x4x4_4b  ", waves_per_simd, NACC, ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4, flop / (ms * 1e-3) * 1e-12);
}

int main()
{
	double *out;
	hipMalloc(&out, sizeof(double) * 256 * 8 * 256 * 8);
	for(int w = 1; w <= 8; w *= 2) {
		run<1, 0>(w, out);
		run<4, 0>(w, out);
		run<16, 0>(w, out);
	}
	for(int w = 1; w <= 8; w *= 2) {
		run<1, 1>(w, out);
		run<8, 1>(w, out);
	}
	return 0;
}
