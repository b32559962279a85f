// Looks for the one-time host-side hiccup seen in loops of sub-millisecond solves: enqueues many
// tiny kernels on three streams without ever synchronizing per launch and reports every launch call that took > 2 ms
// together with the number of launches issued before it. Optional argument: a number of warm-up launches issued (and
// synchronized) first -- if the hiccup is a pool of the runtime growing, it moves into the warm-up.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/launch_stall tools/launch_stall.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
__global__ void tiny(int *p) { if(threadIdx.x == 0 && p) p[0] = 1; }
int main(int argc, char **argv)
{
	const long n_warm = argc > 1 ? atol(argv[1]) : 0, n = argc > 2 ? atol(argv[2]) : 200000;
	hipStream_t s[3];
	for(int i = 0; i < 3; ++ i) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
	int *d; (void)hipMalloc(&d, 64);
	for(long i = 0; i < n_warm; ++ i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i % 3], d);
	(void)hipDeviceSynchronize();
	typedef std::chrono::steady_clock clk;
	double worst = 0; long n_slow = 0;
	const clk::time_point t0 = clk::now();
	for(long i = 0; i < n; ++ i) {
		const clk::time_point a = clk::now();
		hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i % 3], d);
		const double ms = std::chrono::duration<double, std::milli>(clk::now() - a).count();
		if(ms > 2.0) { printf("launch %ld took %.1f ms\n", i, ms); ++ n_slow; }
		if(ms > worst) worst = ms;
		if((i % 200) == 199) (void)hipStreamSynchronize(s[0]); // one solve's worth of launches, then the status fetch
	}
	(void)hipDeviceSynchronize();
	printf("warm %ld, %ld launches in %.1f ms, %ld slow, worst %.2f ms\n", n_warm, n,
		std::chrono::duration<double, std::milli>(clk::now() - t0).count(), n_slow, worst);
	return 0;
}
