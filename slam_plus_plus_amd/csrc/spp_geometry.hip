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

// --------------------------------------------------------------------------------------------------
// SE(3) pose-pose edge CEdgePose3D (vertices [t | axis-angle]). Reference (functional spec):
//   expectation e = C3DJacobians::Absolute_to_Relative(v1, v2): e_t = R1^T (t2 - t1), e_r = log(R1^T R2)
//   Jacobians: forward differences (delta = 1e-9) over v (+) d = Relative_to_Absolute(v, d)
//                                                     include/slam/3DSolverBase.h:1331-1371, :807-850
//   error: [z_t - e_t ; log(R(z_r) R(e_r)^T)]         include/slam/SE3_Types.h:264-286
// Here analytic (R_e = R1^T R2, Jr^-1 = inverse right Jacobian of SO(3) at e_r):
//   d e / d d1 = [ -I   [e_t]x ; 0  -Jr^-1 R_e^T ],   d e / d d2 = [ R_e  0 ; 0  Jr^-1 ]
// agreeing with the reference's difference quotients to their noise (tests/test_gpu_se3_geometry.py).
// One thread per edge; J0, J1: 6 x 6 column-major, r: 6.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_aa(double w, double vx, double vy, double vz, double *a)
{
	if(w < 0) {
		w = -w; vx = -vx; vy = -vy; vz = -vz;
	}
	const double vn = sqrt(vx * vx + vy * vy + vz * vz);
	const double scale = (vn < 1e-12) ? 2.0 : 2.0 * atan2(vn, w) / vn;
	a[0] = vx * scale; a[1] = vy * scale; a[2] = vz * scale;
}

__device__ __forceinline__ void quat_mul(const double *p, const double *q, double *o) // o = p q, (w, x, y, z)
{
	o[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3];
	o[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
	o[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1];
	o[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
}

__global__ __launch_bounds__(256)
void se3_linearize_kernel(int64_t ne, const int32_t *__restrict__ v0, const int32_t *__restrict__ v1,
	const double *__restrict__ poses, const double *__restrict__ meas, double *__restrict__ J0,
	double *__restrict__ J1, double *__restrict__ r)
{
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= ne)
		return;
	const double *p1 = poses + 6 * (int64_t)v0[e], *p2 = poses + 6 * (int64_t)v1[e], *z = meas + 6 * e;
	double R1[9], R2[9], Re[9];
	axis_angle_to_rot(p1 + 3, R1);
	axis_angle_to_rot(p2 + 3, R2);
	const double d0 = p2[0] - p1[0], d1 = p2[1] - p1[1], d2 = p2[2] - p1[2];
	const double et[3] = {R1[0] * d0 + R1[3] * d1 + R1[6] * d2, R1[1] * d0 + R1[4] * d1 + R1[7] * d2,
		R1[2] * d0 + R1[5] * d1 + R1[8] * d2}; // R1^T (t2 - t1)
#pragma unroll
	for(int i = 0; i < 3; ++ i)
#pragma unroll
		for(int j = 0; j < 3; ++ j)
			Re[3 * i + j] = R1[i] * R2[j] + R1[3 + i] * R2[3 + j] + R1[6 + i] * R2[6 + j]; // R1^T R2
	double q1[4], q2[4], qe[4], er[3];
	aa_to_quat(p1 + 3, q1);
	aa_to_quat(p2 + 3, q2);
	q1[1] = -q1[1]; q1[2] = -q1[2]; q1[3] = -q1[3]; // conjugate
	quat_mul(q1, q2, qe);
	quat_to_aa(qe[0], qe[1], qe[2], qe[3], er);
	// error
	double qz[4], qec[4] = {qe[0], -qe[1], -qe[2], -qe[3]}, qr[4], rr[3];
	if(qec[0] < 0) { // the canonical (w >= 0) representative of the expectation, as quat_to_aa / aa_to_quat round-trip
		qec[0] = -qec[0]; qec[1] = -qec[1]; qec[2] = -qec[2]; qec[3] = -qec[3];
	}
	aa_to_quat(z + 3, qz);
	quat_mul(qz, qec, qr);
	quat_to_aa(qr[0], qr[1], qr[2], qr[3], rr);
	double *ro = r + 6 * e;
	ro[0] = z[0] - et[0]; ro[1] = z[1] - et[1]; ro[2] = z[2] - et[2];
	ro[3] = rr[0]; ro[4] = rr[1]; ro[5] = rr[2];
	// Jr^-1(e_r) = I + 1/2 K + c K^2, K = [e_r]x, c = 1/th^2 - (1 + cos th) / (2 th sin th)
	const double th2 = er[0] * er[0] + er[1] * er[1] + er[2] * er[2], th = sqrt(th2);
	double c;
	if(th < 1e-4)
		c = 1.0 / 12.0 + th2 * (1.0 / 720.0);
	else {
		double sn, cs;
		sincos(th, &sn, &cs);
		c = 1.0 / th2 - (1.0 + cs) / (2.0 * th * sn);
	}
	const double x = er[0], y = er[1], zz = er[2];
	double Ji[9] = { // I + K/2 + c K^2 (row-major)
		1 - c * (y * y + zz * zz), -0.5 * zz + c * x * y,      0.5 * y + c * x * zz,
		0.5 * zz + c * x * y,      1 - c * (x * x + zz * zz),  -0.5 * x + c * y * zz,
		-0.5 * y + c * x * zz,     0.5 * x + c * y * zz,       1 - c * (x * x + y * y)};
	double *a = J0 + 36 * e, *b = J1 + 36 * e;
#pragma unroll
	for(int q = 0; q < 36; ++ q) {
		a[q] = 0;
		b[q] = 0;
	}
	// J0 = [ -I  [e_t]x ; 0  -Ji Re^T ]   (column-major: element (row, col) at row + 6 col)
	a[0 + 6 * 0] = -1; a[1 + 6 * 1] = -1; a[2 + 6 * 2] = -1;
	a[0 + 6 * 4] = -et[2]; a[0 + 6 * 5] = et[1];
	a[1 + 6 * 3] = et[2];  a[1 + 6 * 5] = -et[0];
	a[2 + 6 * 3] = -et[1]; a[2 + 6 * 4] = et[0];
#pragma unroll
	for(int i = 0; i < 3; ++ i)
#pragma unroll
		for(int j = 0; j < 3; ++ j) {
			a[(3 + i) + 6 * (3 + j)] = -(Ji[3 * i] * Re[3 * j] + Ji[3 * i + 1] * Re[3 * j + 1] + Ji[3 * i + 2] * Re[3 * j + 2]);
			b[i + 6 * j] = Re[3 * i + j];
			b[(3 + i) + 6 * (3 + j)] = Ji[3 * i + j];
		}
}

// 6D pose (+): t' = t + R dt, R' = R exp(dr) (CVertexPose3D::Operator_Plus, SE3_Types.h:44-47)
__global__ __launch_bounds__(256)
void se3_update_kernel(int64_t nv, double *__restrict__ poses, const double *__restrict__ dx)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= nv)
		return;
	double *p = poses + 6 * i;
	const double *d = dx + 6 * i;
	double R[9], q1[4], q2[4], q[4];
	axis_angle_to_rot(p + 3, R);
	p[0] += R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
	p[1] += R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
	p[2] += R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
	aa_to_quat(p + 3, q1);
	aa_to_quat(d + 3, q2);
	quat_mul(q1, q2, q);
	quat_to_aa(q[0], q[1], q[2], q[3], p + 3);
}

// --------------------------------------------------------------------------------------------------
// Scalars the Levenberg-Marquardt control needs (reference include/slam/NonlinearSolver_Lambda_LM.h):
//   chi2 = sum_e r_e^T Omega_e r_e                                  (f_Error, :1078-1095)
//   alpha0 input = max_e max diag(J_i^T Omega J_i)                  (f_InitialDamping, :151-199)
//   gain denominator = dx . (alpha dx + eta)                        (Aftermath, :204-222)
// Deterministic two-stage reductions (per-workgroup partials in a fixed order, then one workgroup).
// --------------------------------------------------------------------------------------------------
template <int RD>
__global__ __launch_bounds__(256)
void edge_chi2_kernel(int64_t ne, const double *__restrict__ r, const double *__restrict__ Om, double *__restrict__ partial)
{
	__shared__ double red[256];
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	double s = 0;
	if(e < ne) {
		const double *re = r + RD * e, *oe = Om + RD * RD * e;
#pragma unroll
		for(int i = 0; i < RD; ++ i) {
			double t = 0;
#pragma unroll
			for(int j = 0; j < RD; ++ j)
				t += oe[i + RD * j] * re[j];
			s += re[i] * t;
		}
	}
	red[threadIdx.x] = s;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] += red[threadIdx.x + off];
		__syncthreads();
	}
	if(threadIdx.x == 0)
		partial[blockIdx.x] = red[0];
}

template <int RD, int D>
__device__ __forceinline__ double max_hdiag(const double *J, const double *Om)
{
	double m = 0;
#pragma unroll
	for(int c = 0; c < D; ++ c) { // (J^T Omega J)_cc, J: RD x D column-major
		double s = 0;
#pragma unroll
		for(int i = 0; i < RD; ++ i) {
			double t = 0;
#pragma unroll
			for(int j = 0; j < RD; ++ j)
				t += Om[i + RD * j] * J[j + RD * c];
			s += J[i + RD * c] * t;
		}
		m = fmax(m, s);
	}
	return m;
}

template <int RD, int D0, int D1>
__global__ __launch_bounds__(256)
void edge_maxdiag_kernel(int64_t ne, const double *__restrict__ J0, const double *__restrict__ J1,
	const double *__restrict__ Om, double *__restrict__ partial)
{
	__shared__ double red[256];
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	double m = 0;
	if(e < ne)
		m = fmax(max_hdiag<RD, D0>(J0 + RD * D0 * e, Om + RD * RD * e), max_hdiag<RD, D1>(J1 + RD * D1 * e, Om + RD * RD * e));
	red[threadIdx.x] = m;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
		__syncthreads();
	}
	if(threadIdx.x == 0)
		partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256)
void max_partials_kernel(int64_t n, const double *__restrict__ partial, double *__restrict__ out)
{
	__shared__ double red[256];
	double m = 0;
	for(int64_t i = threadIdx.x; i < n; i += 256)
		m = fmax(m, partial[i]);
	red[threadIdx.x] = m;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
		__syncthreads();
	}
	if(threadIdx.x == 0)
		out[0] = red[0];
}

__global__ __launch_bounds__(256)
void gain_partial_kernel(int64_t n, const double *__restrict__ dx, const double *__restrict__ rhs, double alpha,
	double *__restrict__ partial)
{
	__shared__ double red[256];
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	red[threadIdx.x] = (i < n) ? dx[i] * (alpha * dx[i] + rhs[i]) : 0.0;
	__syncthreads();
	for(int off = 128; off > 0; off >>= 1) {
		if((int)threadIdx.x < off)
			red[threadIdx.x] += red[threadIdx.x + off];
		__syncthreads();
	}
	if(threadIdx.x == 0)
		partial[blockIdx.x] = red[0];
}

static double fetch_scalar(spp_ctx *ctx)
{
	double h = 0;
	SPP_HIP_CHECK(hipGetLastError());
	SPP_HIP_CHECK(hipMemcpyAsync(&h, ctx->geom_partial.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return h;
}

// ---- robust kernels on the error norm (include/geometry/RobustLoss.h; mix-in include/slam/RobustUtils.h:368-400):
// w_e = kernel(||r_e|| / scale). kind 0: Huber, w = 1 for x <= k, k / x beyond (CHuberLoss::operator (), :100-104)
__global__ __launch_bounds__(256)
void edge_robust_weight_kernel(int64_t ne, int rd, int kind, double scale, double param, const double *__restrict__ r,
	double *__restrict__ w)
{
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= ne)
		return;
	double s = 0;
	for(int i = 0; i < rd; ++ i)
		s += r[e * rd + i] * r[e * rd + i]; // Eigen's norm(): sqrt of the sum of squares, in order
	const double x = sqrt(s) / scale;
	w[e] = (x <= param) ? 1.0 : param / x;
	(void)kind;
}

void edge_robust_weights(spp_ctx *ctx, int64_t ne, int rd, int kind, double scale, double param, const double *d_r, double *d_w)
{
	SPP_REQUIRE(kind == 0, SPP_E_UNSUPPORTED, "robust weights: only the Huber kernel (kind 0) is instantiated");
	SPP_REQUIRE(scale > 0 && param > 0 && rd > 0, SPP_E_BADARG, "robust weights: scale, parameter and residual dimension must be positive");
	if(!ne)
		return;
	hipLaunchKernelGGL(edge_robust_weight_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, ne, rd, kind,
		scale, param, d_r, d_w);
	SPP_HIP_CHECK(hipGetLastError());
}

double edge_chi2(spp_ctx *ctx, int64_t ne, int rd, const double *d_r, const double *d_Om)
{
	if(!ne)
		return 0;
	const int64_t nwg = (ne + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	double *part = ctx->geom_partial.p + 1;
	const dim3 g((unsigned)nwg), b(256);
	if(rd == 2) hipLaunchKernelGGL((edge_chi2_kernel<2>), g, b, 0, ctx->stream, ne, d_r, d_Om, part);
	else if(rd == 3) hipLaunchKernelGGL((edge_chi2_kernel<3>), g, b, 0, ctx->stream, ne, d_r, d_Om, part);
	else if(rd == 6) hipLaunchKernelGGL((edge_chi2_kernel<6>), g, b, 0, ctx->stream, ne, d_r, d_Om, part);
	else throw Error(SPP_E_UNSUPPORTED, "chi2: residual dimension must be 2, 3 or 6");
	hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, part, ctx->geom_partial.p);
	return fetch_scalar(ctx);
}

double edge_hessian_maxdiag(spp_ctx *ctx, int64_t ne, int rd, int d0, int d1, const double *d_J0, const double *d_J1,
	const double *d_Om)
{
	if(!ne)
		return 0;
	const int64_t nwg = (ne + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	double *part = ctx->geom_partial.p + 1;
	const dim3 g((unsigned)nwg), b(256);
	if(rd == 2 && d0 == 6 && d1 == 3) hipLaunchKernelGGL((edge_maxdiag_kernel<2, 6, 3>), g, b, 0, ctx->stream, ne, d_J0, d_J1, d_Om, part);
	else if(rd == 3 && d0 == 3 && d1 == 3) hipLaunchKernelGGL((edge_maxdiag_kernel<3, 3, 3>), g, b, 0, ctx->stream, ne, d_J0, d_J1, d_Om, part);
	else if(rd == 6 && d0 == 6 && d1 == 6) hipLaunchKernelGGL((edge_maxdiag_kernel<6, 6, 6>), g, b, 0, ctx->stream, ne, d_J0, d_J1, d_Om, part);
	else throw Error(SPP_E_UNSUPPORTED, "max Hessian diagonal: edge group must be (2,6,3), (3,3,3) or (6,6,6)");
	hipLaunchKernelGGL(max_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, part, ctx->geom_partial.p);
	return fetch_scalar(ctx);
}

double lm_gain_denominator(spp_ctx *ctx, int64_t n, const double *d_dx, const double *d_rhs, double alpha)
{
	if(!n)
		return 0;
	const int64_t nwg = (n + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	double *part = ctx->geom_partial.p + 1;
	hipLaunchKernelGGL(gain_partial_kernel, dim3((unsigned)nwg), dim3(256), 0, ctx->stream, n, d_dx, d_rhs, alpha, part);
	hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, part, ctx->geom_partial.p);
	return fetch_scalar(ctx);
}

void se3_linearize(spp_ctx *ctx, int64_t ne, const int32_t *d_v0, const int32_t *d_v1, const double *d_poses,
	const double *d_meas, double *d_J0, double *d_J1, double *d_r)
{
	if(!ne)
		return;
	hipLaunchKernelGGL(se3_linearize_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream,
		ne, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r);
	SPP_HIP_CHECK(hipGetLastError());
}

double se3_update(spp_ctx *ctx, int64_t nv, double *d_poses, const double *d_dx, bool apply)
{
	if(!nv)
		return 0;
	const int64_t n = 6 * nv, nwg = (n + 255) / 256;
	ctx->geom_partial.reserve((size_t)nwg + 1);
	hipLaunchKernelGGL(norm2_partial_kernel, dim3((unsigned)nwg), dim3(256), 0, ctx->stream, n, d_dx, ctx->geom_partial.p + 1);
	hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ctx->stream, nwg, ctx->geom_partial.p + 1, ctx->geom_partial.p);
	if(apply)
		hipLaunchKernelGGL(se3_update_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, ctx->stream, nv, d_poses, d_dx);
	SPP_HIP_CHECK(hipGetLastError());
	double h = 0;
	SPP_HIP_CHECK(hipMemcpyAsync(&h, ctx->geom_partial.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	SPP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return h;
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
