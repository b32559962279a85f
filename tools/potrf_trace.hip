// Phase-level cycle stamps of potrf_diag_kernel (debug harness, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSPP_POTRF_TRACE -I slam_plus_plus_amd/csrc tools/potrf_trace.hip \
//       slam_plus_plus_amd/csrc/spp_api.cpp ... (see tools/build_potrf_trace.sh)
#include "../slam_plus_plus_amd/csrc/spp_dense.hip"
#include <stdio.h>
#include <vector>
#include <random>
#include <math.h>
#include <algorithm>

// the streamed form of the tail kernel's diagonal workgroup: no inverse in the loop, row tiles published per panel
// (mode 1) or not (mode 0)
__global__ __launch_bounds__(spp::POTRF_THREADS)
void potrf_streamed_kernel(double *Ablk, int64_t ld, int n_valid, double *tinv, int *info, int *flag, double *dbuf, int publish)
{
	extern __shared__ double sm[];
	spp::PotrfPub pb;
	if(publish) {
		pb.flag = flag;
		pb.base = 16;
		pb.dbuf = dbuf;
	}
	spp::potrf_diag_body<false, 1, 2, false>(Ablk, ld, n_valid, 0, tinv, info, 0, sm, pb);
}

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
	{
		int *dF; double *dD;
		hipMalloc(&dF, 64); hipMalloc(&dD, 8 * 256 * 8);
		hipFuncSetAttribute((const void*)potrf_streamed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS_DOUBLES_INV2 * (int)sizeof(double));
		for(int publish = 0; publish < 2; ++ publish) {
			for(int it = 0; it < 3; ++ it) {
				hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
				hipMemset(dI, 0, 16);
				hipLaunchKernelGGL(potrf_streamed_kernel, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_DOUBLES_INV2 * sizeof(double), 0, dA, (int64_t)n, n, dT, dI, dF, dD, publish);
				hipDeviceSynchronize();
			}
			long long u[64];
			hipMemcpyFromSymbol(u, HIP_SYMBOL(spp_potrf_trace), sizeof(u));
			auto e = [&](int a, int b) { return (double)(u[b] - u[a]); };
			printf("streamed form (HALF = 2), publish %d: total %.0f cycles; load %.0f; loop %.0f; after the loop (inverse) %.0f\n", publish, e(0, 52), e(0, 2), e(2, 51), e(51, 52));
			for(int J = 0; J < 8; ++ J)
				printf("  J=%d  B %.0f  C(wave0) %.0f  C(wave1) %.0f  panel total %.0f\n", J, e(3 + 6 * J, 4 + 6 * J), e(5 + 6 * J, 7 + 6 * J), e(6 + 6 * J, 8 + 6 * J),
					J < 7 ? e(3 + 6 * J, 3 + 6 * (J + 1)) : e(3 + 6 * J, 51));
		}
		// (dA / dT now hold the streamed form's R and whole inverse: the check below completes a two-halves inverse, so restore the default kernel's results)
		hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
		hipMemset(dI, 0, 16);
		hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_DOUBLES * sizeof(double), 0, dA, (int64_t)n, n, 0, dT, dI, (int64_t)0);
		hipDeviceSynchronize();
	}
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
