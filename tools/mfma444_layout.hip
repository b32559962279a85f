// Operand / result lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950, found empirically: lane la feeds A = 1 (all others 0),
// lane lb feeds B = 1; the lanes whose D comes out 1 tell which (block, i, k) / (block, k, j) the two lanes hold.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *mask)
{
	const int lane = threadIdx.x;
	for(int la = 0; la < 64; ++ la)
		for(int lb = 0; lb < 64; ++ lb) {
			const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
			const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
			const unsigned long long m = __builtin_amdgcn_ballot_w64(d != 0.0);
			if(lane == 0)
				mask[la * 64 + lb] = m;
		}
}
int main()
{
	unsigned long long *d, h[4096];
	hipMalloc(&d, sizeof(h));
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	for(int la = 0; la < 64; ++ la) {
		printf("A lane %2d pairs with B lanes -> D lane:", la);
		for(int lb = 0; lb < 64; ++ lb)
			if(h[la * 64 + lb])
				printf("  %d->%d", lb, __builtin_ctzll(h[la * 64 + lb]));
		printf("\n");
	}
	return 0;
}
