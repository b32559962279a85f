// spp_geometry.hip -- on-device edge linearization for 2D pose graphs (the
// CEdgePose2D part): Jacobians, expectation and error of the relative-pose edge, the vertex update
// x <- x (+) dx and ||dx||, so that a whole Gauss-Newton iteration of a 2D pose graph stays in HBM.
// gfx950 only.
//
// Reference (functional spec, nothing is ported):
//   C2DJacobians::Absolute_to_Relative with Jacobians   include/slam/2DSolverBase.h:373-418
//   C2DJacobians::f_ClampAngle_2Pi / f_ClampAngularError_2Pi  :44-94
//   CEdgePose2D::Calculate_Jacobians_Expectation_Error  include/slam/SE2_Types.h (error = z - h(x))
//   CVertexPose2D::Operator_Plus                        include/slam/SE2_Types.h:70-74
// The reference's 2D Jacobians are analytic, so the device values agree with it to rounding (the BA
// and SE(3) edges use forward differences with delta = 1e-9 there: SURVEY 8f rank 2 explains why
// those need a Delta-x-level tolerance instead).
//
// Layout = what spp_assemble_device consumes: J0, J1: ne x (3 x 3) column-major, r: ne x 3.
// One thread per edge: 2 x 24 B of gathered poses + 24 B measurement in, 168 B out; HBM-bound.

#include "spp_internal.h"
#include <math.h>

namespace spp {

__device__ __forceinline__ double clamp_angle_2pi(double a)
{
	return fmod(a, 6.283185307179586476925286766559);
}

__device__ __forceinline__ double clamp_angular_error_2pi(double e)
{
	e = clamp_angle_2pi(e);
	const double a = e - 6.283185307179586476925286766559, b = e + 6.283185307179586476925286766559;
	double m = e;
	if(fabs(a) < fabs(m)) m = a;
	if(fabs(b) < fabs(m)) m = b;
	return m;
}

__global__ __launch_bounds__(256)
void se2_linearize_kernel(int64_t ne, const int32_t *__restrict__ v0, const int32_t *__restrict__ v1,
	const double *__restrict__ poses, const double *__restrict__ meas, double *__restrict__ J0,
	double *__restrict__ J1, double *__restrict__ r)
{
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= ne)
		return;
	const double *p1 = poses + 3 * (int64_t)v0[e], *p2 = poses + 3 * (int64_t)v1[e];
	const double p1e = p1[0], p1n = p1[1], p1a = p1[2];
	const double de = p2[0] - p1e, dn = p2[1] - p1n;
	double s, c;
	sincos(p1a, &s, &c);
	// expectation h(x): the second pose in the frame of the first
	const double hf = c * de + s * dn, hl = -s * de + c * dn, ha = clamp_angle_2pi(p2[2] - p1a);
	const double *z = meas + 3 * e;
	r[3 * e + 0] = z[0] - hf;
	r[3 * e + 1] = z[1] - hl;
	r[3 * e + 2] = clamp_angular_error_2pi(z[2] - ha);
	// d h / d pose1 (3 x 3, column-major)
	double *a = J0 + 9 * e;
	a[0] = -c;  a[1] = s;   a[2] = 0;
	a[3] = -s;  a[4] = -c;  a[5] = 0;
	a[6] = -s * de + c * dn;
	a[7] = -c * de - s * dn;
	a[8] = -1;
	// d h / d pose2
	double *b = J1 + 9 * e;
	b[0] = c;   b[1] = -s;  b[2] = 0;
	b[3] = s;   b[4] = c;   b[5] = 0;
	b[6] = 0;   b[7] = 0;   b[8] = 1;
}

// x <- x (+) dx for 2D poses (add, clamp the angle); per-workgroup partial sums of dx^2 in a FIXED order
__global__ __launch_bounds__(256)
void se2_update_kernel(int64_t nv, double *__restrict__ poses, const double *__restrict__ dx, int apply,
	double *__restrict__ partial)
{
	__shared__ double red[256];
	const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	double s = 0;
	if(v < nv) {
		const double d0 = dx[3 * v], d1 = dx[3 * v + 1], d2 = dx[3 * v + 2];
		s = d0 * d0 + d1 * d1 + d2 * d2;
		if(apply) {
			poses[3 * v] += d0;
			poses[3 * v + 1] += d1;
			poses[3 * v + 2] = clamp_angle_2pi(poses[3 * v + 2] + d2);
		}
	}
	red[threadIdx.x] = s;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] += red[threadIdx.x + off];
		__syncthreads();
	}
	if(threadIdx.x == 0 && partial)
		partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256)
void sum_partials_kernel(int64_t n, const double *__restrict__ partial, double *__restrict__ out)
{
	__shared__ double red[256];
	double s = 0;
	for(int64_t i = threadIdx.x; i < n; i += 256) // fixed assignment, fixed order: bit-reproducible
		s += partial[i];
	red[threadIdx.x] = s;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] += red[threadIdx.x + off];
		__syncthreads();
	}
	if(threadIdx.x == 0)
		out[0] = red[0];
}

void se2_linearize(spp_ctx *ctx, int64_t ne, const int32_t *d_v0, const int32_t *d_v1, const double *d_poses,
	const double *d_meas, double *d_J0, double *d_J1, double *d_r)
{
	if(!ne)
		return;
	hipLaunchKernelGGL(se2_linearize_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream,
		ne, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r);
	SPP_HIP_CHECK(hipGetLastError());
}

// returns ||dx||^2 (synchronizes the stream: the caller needs the value for the stopping test, as the
// reference does, NonlinearSolver_Lambda.h:638-650)
double se2_update(spp_ctx *ctx, int64_t nv, double *d_poses, const double *d_dx, bool apply)
{
	if(!nv)
		return 0;
	const int64_t nwg = (nv + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	hipLaunchKernelGGL(se2_update_kernel, dim3((unsigned)nwg), dim3(256), 0, ctx->stream, nv, d_poses, d_dx,
		apply ? 1 : 0, ctx->geom_partial.p + 1);
	hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, ctx->geom_partial.p + 1,
		ctx->geom_partial.p);
	SPP_HIP_CHECK(hipGetLastError());
	double h = 0;
	SPP_HIP_CHECK(hipMemcpyAsync(&h, ctx->geom_partial.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return h;
}

} // namespace spp
