// Phase-level cycle stamps of potrf_diag_kernel (debug harness, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSPP_POTRF_TRACE -I slam_plus_plus_amd/csrc tools/potrf_trace.hip \
//       slam_plus_plus_amd/csrc/spp_api.cpp ... (see tools/build_potrf_trace.sh)
#include "../slam_plus_plus_amd/csrc/spp_dense.hip"
#include <stdio.h>
#include <vector>
#include <random>
#include <math.h>
#include <algorithm>

int main()
{
	using namespace spp;
	const int n = 128;
	std::vector<double> A(n * n), M(n * n);
	std::mt19937_64 g(1);
	std::normal_distribution<double> nd;
	for(auto &v : M) v = nd(g);
	for(int i = 0; i < n; ++ i)
		for(int j = 0; j < n; ++ j) {
			double s = 0;
			for(int k = 0; k < n; ++ k) s += M[i + k * n] * M[j + k * n];
			A[i + j * n] = s / n + (i == j ? 2.0 : 0.0);
		}
	double *dA, *dT; int *dI;
	hipMalloc(&dA, n * n * 8); hipMalloc(&dT, n * n * 8); hipMalloc(&dI, 16);
	hipFuncSetAttribute((const void*)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_DOUBLES * (int)sizeof(double));
	for(int it = 0; it < 3; ++ it) {
		hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
		hipMemset(dI, 0, 16);
		hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_DOUBLES * sizeof(double), 0, dA, (int64_t)n, n, 0, dT, dI, (int64_t)0);
		hipDeviceSynchronize();
	}
	long long t[64];
	hipMemcpyFromSymbol(t, HIP_SYMBOL(spp_potrf_trace), sizeof(t));
	auto d = [&](int a, int b) { return (double)(t[b] - t[a]); };
	printf("total %.0f cycles; load %.0f; first diag %.0f; writeback %.0f\n", d(0, 52), d(0, 1), d(1, 2), d(51, 52));
	for(int J = 0; J < 8; ++ J)
		printf("J=%d  B %.0f  barrier %.0f  C(wave0: upd+factor) %.0f  C(wave1) %.0f  panel total %.0f\n", J,
			d(3 + 6 * J, 4 + 6 * J), d(4 + 6 * J, 5 + 6 * J), d(5 + 6 * J, 7 + 6 * J), d(6 + 6 * J, 8 + 6 * J),
			J < 7 ? d(3 + 6 * J, 3 + 6 * (J + 1)) : d(3 + 6 * J, 51));
	// the results of the last launch against the input: |R^T R - A| and |R Tinv - I| (upper triangles)
	std::vector<double> R(n * n), Ti(n * n);
	hipMemcpy(R.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
	hipMemcpy(Ti.data(), dT, n * n * 8, hipMemcpyDeviceToHost);
	{ // the kernel leaves [T0 R01; 0 T1]: complete it to the whole inverse, -T0 R01 T1 between the halves
		std::vector<double> N(64 * 64);
		for(int i = 0; i < 64; ++ i)
			for(int j = 0; j < 64; ++ j) {
				double s = 0;
				for(int k = 0; k < 64; ++ k) s += Ti[i + (64 + k) * n] * Ti[64 + k + (64 + j) * n];
				N[i + j * 64] = s;
			}
		for(int i = 0; i < 64; ++ i)
			for(int j = 0; j < 64; ++ j) {
				double s = 0;
				for(int k = 0; k < 64; ++ k) s += Ti[i + k * n] * N[k + j * 64];
				Ti[i + (64 + j) * n] = -s;
			}
	}
	double e1 = 0, e2 = 0, e3 = 0;
	for(int i = 0; i < n; ++ i)
		for(int j = i; j < n; ++ j) {
			double s = 0, u = 0;
			for(int k = 0; k <= i; ++ k) s += R[k + i * n] * R[k + j * n];
			for(int k = i; k <= j; ++ k) u += R[i + k * n] * Ti[k + j * n];
			e1 = std::max(e1, fabs(s - A[i + j * n]));
			e2 = std::max(e2, fabs(u - (i == j ? 1.0 : 0.0)));
		}
	for(int i = 0; i < n; ++ i)
		for(int j = 0; j < i; ++ j)
			e3 = std::max(e3, fabs(Ti[i + j * n]));
	printf("check: |R^T R - A| %.3g   |R Tinv - I| %.3g   |tril(Tinv, -1)| %.3g\n", e1, e2, e3);
	return 0;
}
