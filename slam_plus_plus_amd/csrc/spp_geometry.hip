// spp_geometry.hip -- on-device edge linearization: Jacobians, expectation and error
// of the 2D relative-pose edge (CEdgePose2D) and of the BA projection edge (CEdgeP2C3D), the vertex
// updates x <- x (+) dx and ||dx||, so that a whole Gauss-Newton / LM iteration stays in HBM.
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

// --------------------------------------------------------------------------------------------------
// Bundle adjustment: projection edge CEdgeP2C3D (camera 6D pose [t | axis-angle], world -> camera,
// + 5 constant intrinsics fx fy cx cy k; point XYZ). Reference (functional spec):
//   CBAJacobians::Project_P2C               include/slam/BASolverBase.h:260-325 (model), :559-620 (Jacobians)
//     x = R(aa) X + t ; d = (fx x/z, fy y/z) ; k' = k / ((fx + fy) / 2) ; uv = c + (1 + |d|^2 k') d
//   its Jacobians are FORWARD DIFFERENCES (delta = 1e-9) over the camera increment
//     cam (+) delta = C3DJacobians::Relative_to_Absolute(cam, delta): t' = t + R dt, R' = R exp(dr)
//     (include/slam/3DSolverBase.h:807-850) and over an additive point increment.
// Here the same derivatives are ANALYTIC:  d uv / d x = ((1 + r2 k') I + 2 k' d d^T) [fx/z 0 -fx x/z^2; 0 fy/z -fy y/z^2]
//   J_cam = d uv/d x [ R | -R [X]x ],  J_pt = d uv/d x R,   r = z - uv.
// They agree with the reference's difference quotients to the quotients' own noise (~1e-7 relative:
// tests/test_gpu_ba_geometry.py compares with golden vectors produced by the reference).
// One thread per observation: 11 + 3 gathered doubles + 16 B in, 160 B out (J0 2x6, J1 2x3 column-major, r).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void axis_angle_to_rot(const double *a, double *R) // row-major 3 x 3
{
	const double x = a[0], y = a[1], z = a[2], th2 = x * x + y * y + z * z, th = sqrt(th2);
	double A, B; // sin(th)/th, (1 - cos(th))/th^2
	if(th < 1e-6) {
		A = 1.0 - th2 * (1.0 / 6.0);
		B = 0.5 - th2 * (1.0 / 24.0);
	} else {
		double s, c;
		sincos(th, &s, &c);
		A = s / th;
		const double sh = sin(0.5 * th);
		B = 2.0 * sh * sh / th2;
	}
	R[0] = 1 - B * (y * y + z * z); R[1] = B * x * y - A * z;       R[2] = B * x * z + A * y;
	R[3] = B * x * y + A * z;       R[4] = 1 - B * (x * x + z * z); R[5] = B * y * z - A * x;
	R[6] = B * x * z - A * y;       R[7] = B * y * z + A * x;       R[8] = 1 - B * (x * x + y * y);
}

__global__ __launch_bounds__(256)
void ba_linearize_kernel(int64_t no, const int32_t *__restrict__ cam_of, const int32_t *__restrict__ pt_of,
	const double *__restrict__ cams, const double *__restrict__ intr, const double *__restrict__ pts,
	const double *__restrict__ meas, double *__restrict__ J0, double *__restrict__ J1, double *__restrict__ r)
{
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= no)
		return;
	const double *cam = cams + 6 * (int64_t)cam_of[e], *in = intr + 5 * (int64_t)cam_of[e], *X = pts + 3 * (int64_t)pt_of[e];
	double R[9];
	axis_angle_to_rot(cam + 3, R);
	const double X0 = X[0], X1 = X[1], X2 = X[2];
	const double x = R[0] * X0 + R[1] * X1 + R[2] * X2 + cam[0];
	const double y = R[3] * X0 + R[4] * X1 + R[5] * X2 + cam[1];
	const double z = R[6] * X0 + R[7] * X1 + R[8] * X2 + cam[2];
	const double fx = in[0], fy = in[1], k = in[4] / (0.5 * (fx + fy));
	const double iz = 1.0 / z, d0 = fx * x * iz, d1 = fy * y * iz, r2 = d0 * d0 + d1 * d1, g = 1.0 + r2 * k;
	r[2 * e] = meas[2 * e] - (in[2] + g * d0);
	r[2 * e + 1] = meas[2 * e + 1] - (in[3] + g * d1);
	// d uv / d x (2 x 3): D * Jd
	const double D00 = g + 2 * k * d0 * d0, D01 = 2 * k * d0 * d1, D11 = g + 2 * k * d1 * d1;
	const double a0 = fx * iz, a2 = -fx * x * iz * iz, b1 = fy * iz, b2 = -fy * y * iz * iz; // Jd = [a0 0 a2; 0 b1 b2]
	const double P[6] = {D00 * a0, D01 * b1, D00 * a2 + D01 * b2,   // row 0
	                     D01 * a0, D11 * b1, D01 * a2 + D11 * b2};  // row 1
	// PR = P R (2 x 3) = d uv / d dt = d uv / d X
	double PR[6];
#pragma unroll
	for(int i = 0; i < 2; ++ i)
#pragma unroll
		for(int j = 0; j < 3; ++ j)
			PR[3 * i + j] = P[3 * i] * R[j] + P[3 * i + 1] * R[3 + j] + P[3 * i + 2] * R[6 + j];
	double *a = J0 + 12 * e, *b = J1 + 6 * e;
#pragma unroll
	for(int j = 0; j < 3; ++ j) {
		a[2 * j] = PR[j];
		a[2 * j + 1] = PR[3 + j];
		b[2 * j] = PR[j];
		b[2 * j + 1] = PR[3 + j];
	}
	// d uv / d dr = -PR [X]x : columns (PR x X) component-wise: -PR * [X]x = [PR_1 X2 - PR_2 X1, PR_2 X0 - PR_0 X2, PR_0 X1 - PR_1 X0] * (-1) ...
#pragma unroll
	for(int i = 0; i < 2; ++ i) {
		const double p0 = PR[3 * i], p1 = PR[3 * i + 1], p2 = PR[3 * i + 2];
		// -(p^T [X]x) with [X]x = [0 -X2 X1; X2 0 -X0; -X1 X0 0]: p^T [X]x = (p1 X2 - p2 X1, p2 X0 - p0 X2, p0 X1 - p1 X0)
		a[6 + i] = -(p1 * X2 - p2 * X1);
		a[8 + i] = -(p2 * X0 - p0 * X2);
		a[10 + i] = -(p0 * X1 - p1 * X0);
	}
}

// camera (+): t' = t + R dt, R' = R exp(dr) through unit quaternions with w >= 0 (the reference's
// AxisAngle_to_Quat / Quat_to_AxisAngle, 3DSolverBase.h:477-502,557+); one thread per camera
__device__ __forceinline__ void aa_to_quat(const double *a, double *q) // q = (w, x, y, z)
{
	const double th = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
	double c, s_over;
	if(th < 1e-12) {
		c = 1.0;
		s_over = 0.5;
	} else {
		double sh;
		sincos(0.5 * th, &sh, &c);
		s_over = sh / th;
		if(c < 0) {
			c = -c;
			s_over = -s_over;
		}
	}
	q[0] = c; q[1] = a[0] * s_over; q[2] = a[1] * s_over; q[3] = a[2] * s_over;
}

__global__ __launch_bounds__(256)
void ba_update_cams_kernel(int64_t nc, double *__restrict__ cams, const int64_t *__restrict__ dxoff, const double *__restrict__ dx)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= nc)
		return;
	double *cam = cams + 6 * i;
	const double *d = dx + dxoff[i];
	double R[9], q1[4], q2[4];
	axis_angle_to_rot(cam + 3, R);
	cam[0] += R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
	cam[1] += R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
	cam[2] += R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
	aa_to_quat(cam + 3, q1);
	aa_to_quat(d + 3, q2);
	double w = q1[0] * q2[0] - q1[1] * q2[1] - q1[2] * q2[2] - q1[3] * q2[3];
	double vx = q1[0] * q2[1] + q1[1] * q2[0] + q1[2] * q2[3] - q1[3] * q2[2];
	double vy = q1[0] * q2[2] - q1[1] * q2[3] + q1[2] * q2[0] + q1[3] * q2[1];
	double vz = q1[0] * q2[3] + q1[1] * q2[2] - q1[2] * q2[1] + q1[3] * q2[0];
	if(w < 0) {
		w = -w; vx = -vx; vy = -vy; vz = -vz;
	}
	const double vn = sqrt(vx * vx + vy * vy + vz * vz);
	const double scale = (vn < 1e-12) ? 2.0 : 2.0 * atan2(vn, w) / vn;
	cam[3] = vx * scale;
	cam[4] = vy * scale;
	cam[5] = vz * scale;
}

__global__ __launch_bounds__(256)
void ba_update_points_kernel(int64_t np, double *__restrict__ pts, const int64_t *__restrict__ dxoff, const double *__restrict__ dx)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= np)
		return;
	const double *d = dx + dxoff[i];
	pts[3 * i] += d[0];
	pts[3 * i + 1] += d[1];
	pts[3 * i + 2] += d[2];
}

__global__ __launch_bounds__(256)
void norm2_partial_kernel(int64_t n, const double *__restrict__ v, double *__restrict__ partial)
{
	__shared__ double red[256];
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	red[threadIdx.x] = (i < n) ? v[i] * v[i] : 0.0;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] += red[threadIdx.x + off];
		__syncthreads();
	}
	if(threadIdx.x == 0)
		partial[blockIdx.x] = red[0];
}

void ba_linearize(spp_ctx *ctx, int64_t no, const int32_t *d_cam_of, const int32_t *d_pt_of, const double *d_cams,
	const double *d_intr, const double *d_pts, const double *d_meas, double *d_J0, double *d_J1, double *d_r)
{
	if(!no)
		return;
	hipLaunchKernelGGL(ba_linearize_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, ctx->stream,
		no, d_cam_of, d_pt_of, d_cams, d_intr, d_pts, d_meas, d_J0, d_J1, d_r);
	SPP_HIP_CHECK(hipGetLastError());
}

double ba_update(spp_ctx *ctx, int64_t nc, double *d_cams, const int64_t *d_cam_dxoff, int64_t np, double *d_pts,
	const int64_t *d_pt_dxoff, const double *d_dx, int64_t n_dx, bool apply)
{
	const int64_t nwg = (n_dx + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	if(n_dx) {
		hipLaunchKernelGGL(norm2_partial_kernel, dim3((unsigned)nwg), dim3(256), 0, ctx->stream, n_dx, d_dx, ctx->geom_partial.p + 1);
		hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, ctx->geom_partial.p + 1, ctx->geom_partial.p);
	}
	if(apply && nc)
		hipLaunchKernelGGL(ba_update_cams_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, nc, d_cams, d_cam_dxoff, d_dx);
	if(apply && np)
		hipLaunchKernelGGL(ba_update_points_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, np, d_pts, d_pt_dxoff, d_dx);
	SPP_HIP_CHECK(hipGetLastError());
	double h = 0;
	if(n_dx)
		SPP_HIP_CHECK(hipMemcpyAsync(&h, ctx->geom_partial.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return h;
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
