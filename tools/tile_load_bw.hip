// How fast can ONE workgroup pull 128 x 128 double tiles out of a column-major matrix (leading dimension ld)?
// The backward substitution's chain workgroup lives on exactly that. Patterns (256 threads = 4 waves unless noted):
//   0  8 bytes per lane, lane -> row (tid & 127), 64 columns per thread        (two 512-byte segments per wave load)
//   1  8 bytes per lane, 2 threads per row (tid / 2), 64 columns per thread     (as trsv_back_chain2 TPR = 2)
//   2  16 bytes per lane: a wave load = one whole tile column (1 KB contiguous), wave w takes columns w, w + 4, ...
//   3  as 2 on a tile repacked contiguously (ld = 128)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/tile_load_bw tools/tile_load_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void pull(const double *A, long ld, int ntile, long tile_step, double *out, long long *cyc)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double acc = 0;
	const long long t0 = wall_clock64(); // 100 MHz
	for(int t = 0; t < ntile; ++ t) {
		const double *T = A + t * tile_step;
		if(MODE == 0) {
			const int r = tid & 127, h = tid >> 7;
			double v[64];
#pragma unroll
			for(int c = 0; c < 64; ++ c)
				v[c] = T[r + (long)(64 * h + c) * ld];
#pragma unroll
			for(int c = 0; c < 64; ++ c)
				acc += v[c];
		} else if(MODE == 1) {
			const int r = tid >> 1, h = tid & 1;
			double v[64];
#pragma unroll
			for(int c = 0; c < 64; ++ c)
				v[c] = T[r + (long)(64 * h + c) * ld];
#pragma unroll
			for(int c = 0; c < 64; ++ c)
				acc += v[c];
		} else {
			double2 v[32];
#pragma unroll
			for(int c = 0; c < 32; ++ c)
				v[c] = *(const double2*)(T + 2 * lane + (long)(wave + 4 * c) * ld);
#pragma unroll
			for(int c = 0; c < 32; ++ c)
				acc += v[c].x + v[c].y;
		}
	}
	const long long t1 = wall_clock64();
	out[tid] = acc;
	if(tid == 0)
		*cyc = t1 - t0;
}

int main()
{
	const long ld = 5248, n = 5248;
	double *A, *out;
	long long *cyc;
	hipMalloc(&A, ld * n * 8);
	hipMemset(A, 0, ld * n * 8);
	hipMalloc(&out, 256 * 8);
	hipMalloc(&cyc, 8);
	const int ntile = 40;
	const long step = 128 + 128 * ld; // tile (b, b + 1) -> (b + 1, b + 2)
	for(int rep = 0; rep < 2; ++ rep)
		for(int mode = 0; mode < 4; ++ mode) {
			const long l = mode == 3 ? 128 : ld, st = mode == 3 ? 128 * 128 : step;
			if(mode == 0) hipLaunchKernelGGL(pull<0>, dim3(1), dim3(256), 0, 0, A + 128 * ld, l, ntile, st, out, cyc);
			if(mode == 1) hipLaunchKernelGGL(pull<1>, dim3(1), dim3(256), 0, 0, A + 128 * ld, l, ntile, st, out, cyc);
			if(mode >= 2) hipLaunchKernelGGL(pull<2>, dim3(1), dim3(256), 0, 0, A + 128 * ld, l, ntile, st, out, cyc);
			hipDeviceSynchronize();
			long long c;
			hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
			// wall_clock64 ticks at 100 MHz
			printf("mode %d: %.2f us per 128 KB tile = %.1f GB/s\n", mode, c * 0.01 / ntile, 131072.0 / (c * 0.01 / ntile) * 1e-3);
		}
	return 0;
}
