/*
 * oracle/dropin_driver.cpp -- TEST INFRASTRUCTURE. Proves the drop-in boundary end to end:
 * the reference's OWN nonlinear solver (CNonlinearSolver_Lambda::Optimize,
 * include/slam/NonlinearSolver_Lambda.h:476-883) is instantiated twice on the same generated 2D
 * pose graph -- once with the reference's CLinearSolver_UberBlock, once with CLinearSolver_HIP
 * (include/spp_adapter.h -> libspp_hip.so -> MI355X) -- and the optimized vertex states are
 * compared. Built by oracle/Makefile into oracle/_ref/dropin_driver (it contains reference code,
 * so it lives next to libspp_ref.so and never enters git).
 *
 * A second mode does the same for bundle adjustment with the reference's LEVENBERG-MARQUARDT solver
 * (CNonlinearSolver_Lambda_LM, include/slam/NonlinearSolver_Lambda_LM.h:796-1160 -- what slam_app
 * silently uses for every BA input, src/slam_app/Main.cpp:203-208): reference = UberBlock behind
 * the reference's own CLinearSolver_Schur (-us), ours = CLinearSolver_HIP alone (it eliminates the
 * landmarks itself on the GPU).
 *
 * usage: dropin_driver [n_poses] [n_loop_closures]     ->  "poses .. max_abs_diff <d> .."
 *        dropin_driver ba [n_cams] [n_points]           ->  "ba cams .. max_abs_diff <d> .."
 *        dropin_driver dump n_poses n_loop_closures file -> golden vector of the reference's Gauss-Newton
 *            loop alone (reference linear solver, no GPU): edges, the initial vertex states the reference
 *            derives from them, and the states after Optimize(5, 0.01) (slam_app's defaults,
 *            src/slam_app/Main.cpp:706-707). tools/make_golden_gn.py turns it into tests/golden/*.npz.
 *        dropin_driver badump n file                     -> golden vectors of the reference's BA edge geometry
 *            (no solver, no GPU): n random (camera, intrinsics, point) triples with the expectation and the
 *            two Jacobians of CBAJacobians::Project_P2C (include/slam/BASolverBase.h:559-620, forward
 *            differences over the SE(3) composition C3DJacobians::Relative_to_Absolute), and the result of
 *            that composition for a random 6D increment (the camera (+) of CVertexCam::Operator_Plus).
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <thread>
#include <algorithm>

#include "slam/LinearSolver_UberBlock.h"
#include "slam/ConfigSolvers.h"
#include "slam/SE2_Types.h"
#include "slam/BA_Types.h"
#include "slam/SE3_Types.h"
#include "slam/NonlinearSolver_Lambda_LM.h"
#include "spp_adapter.h"

typedef MakeTypelist(CVertexPose2D) TVertexTypelist;
typedef MakeTypelist(CEdgePose2D) TEdgeTypelist;
typedef CFlatSystem<CVertexPose2D, TVertexTypelist, CEdgePose2D, TEdgeTypelist> CSystemType;

struct TEdge { size_t a, b; Eigen::Vector3d z; };

static unsigned long long g_state = 88172645463325252ull;
static double Rand01() // xorshift: deterministic, no dependence on libc's rand
{
	g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
	return double(g_state >> 11) / 9007199254740992.0;
}
static double RandN()
{
	double u = Rand01(), v = Rand01();
	return sqrt(-2 * log(u + 1e-300)) * cos(6.283185307179586 * v);
}

static void Generate(size_t n_poses, size_t n_loops, std::vector<TEdge> &r_edges)
{
	std::vector<double> x(n_poses), y(n_poses), th(n_poses);
	int heading = 0;
	x[0] = y[0] = th[0] = 0;
	for(size_t i = 1; i < n_poses; ++ i) {
		if(Rand01() < 0.25)
			heading = (heading + ((Rand01() < 0.5)? 1 : 3)) % 4;
		th[i] = heading * 1.5707963267948966;
		x[i] = x[i - 1] + cos(th[i - 1]);
		y[i] = y[i - 1] + sin(th[i - 1]);
	}
	struct L { static TEdge Make(size_t a, size_t b, const std::vector<double> &x, const std::vector<double> &y,
		const std::vector<double> &th) {
			double c = cos(th[a]), s = sin(th[a]), dx = x[b] - x[a], dy = y[b] - y[a];
			TEdge e; e.a = a; e.b = b;
			e.z = Eigen::Vector3d(c * dx + s * dy + 0.03 * RandN(), -s * dx + c * dy + 0.03 * RandN(),
				th[b] - th[a] + 0.01 * RandN());
			return e;
		} };
	for(size_t i = 0; i + 1 < n_poses; ++ i)
		r_edges.push_back(L::Make(i, i + 1, x, y, th));
	for(size_t k = 0, n_tries = 0; k < n_loops && n_tries < 100 * n_loops + 1000; ++ n_tries) {
		size_t a = size_t(Rand01() * n_poses), b = size_t(Rand01() * n_poses);
		if(a + 1 >= b || b >= n_poses)
			continue;
		if(fabs(x[a] - x[b]) + fabs(y[a] - y[b]) > 3)
			continue;
		r_edges.push_back(L::Make(a, b, x, y, th));
		++ k;
	}
}

template <class CLinearSolverType>
static bool Run(const std::vector<TEdge> &r_edges, std::vector<double> &r_state, size_t n_max_iter,
	double f_threshold = 1e-6, std::vector<double> *p_initial_state = 0)
{
	CSystemType system;
	CNonlinearSolver_Lambda<CSystemType, CLinearSolverType> solver(system);
	Eigen::Matrix3d information;
	information << 1111.11, 0, 0, 0, 1111.11, 0, 0, 0, 10000;
	for(size_t i = 0; i < r_edges.size(); ++ i)
		system.r_Add_Edge(CEdgePose2D(r_edges[i].a, r_edges[i].b, r_edges[i].z, information, system));
	if(p_initial_state) {
		p_initial_state->clear();
		for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
			Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
			for(int j = 0; j < v.rows(); ++ j)
				p_initial_state->push_back(v(j));
		}
	}
	solver.Optimize(n_max_iter, f_threshold);
	r_state.clear();
	for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
		Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
		for(int j = 0; j < v.rows(); ++ j)
			r_state.push_back(v(j));
	}
	return true;
}

// ---- bundle adjustment through the reference's LM solver -----------------------------------------
typedef MakeTypelist_Safe((CVertexCam, CVertexXYZ)) TBAVertexTypelist;
typedef MakeTypelist_Safe((CEdgeP2C3D)) TBAEdgeTypelist;
typedef CFlatSystem<CBaseVertex, TBAVertexTypelist, CEdgeP2C3D, TBAEdgeTypelist> CBASystemType;

struct TBAProblem {
	std::vector<Eigen::Matrix<double, 11, 1> > cams; // initial estimates
	std::vector<Eigen::Vector3d> points;
	struct TObs { size_t n_cam, n_pt; Eigen::Vector2d z; };
	std::vector<TObs> obs;
};

static void Generate_BA(size_t n_cams, size_t n_points, TBAProblem &r_p)
{
	std::vector<Eigen::Matrix<double, 6, 1> > true_cams(n_cams);
	Eigen::Matrix<double, 5, 1> intr;
	intr << 500, 500, 0, 0, 0;
	for(size_t i = 0; i < n_cams; ++ i) {
		double th = 6.283185307179586 * i / n_cams;
		Eigen::Vector3d C(10 * cos(th), 10 * sin(th), 0.5 * sin(3 * th));
		Eigen::Vector3d z = -C.normalized(), x = Eigen::Vector3d(0, 0, 1).cross(z).normalized(), y = z.cross(x);
		Eigen::Matrix3d R;
		R.row(0) = x; R.row(1) = y; R.row(2) = z; // world -> camera
		Eigen::AngleAxisd aa(R);
		true_cams[i].head<3>() = -R * C;
		true_cams[i].tail<3>() = aa.axis() * aa.angle();
		Eigen::Matrix<double, 11, 1> est;
		est.head<6>() = true_cams[i];
		for(int k = 0; k < 3; ++ k) {
			est(k) += 0.02 * RandN();
			est(3 + k) += 0.002 * RandN();
		}
		est.tail<5>() = intr;
		r_p.cams.push_back(est);
	}
	for(size_t j = 0; j < n_points; ++ j) {
		Eigen::Vector3d X(4 * Rand01() - 2, 4 * Rand01() - 2, 4 * Rand01() - 2);
		size_t k = 3 + size_t(Rand01() * 4), c0 = size_t(Rand01() * n_cams);
		for(size_t q = 0; q < k && q < n_cams; ++ q) {
			size_t c = (c0 + q * (1 + n_cams / 9)) % n_cams;
			Eigen::Vector2d z;
			CBAJacobians::Project_P2C(true_cams[c], intr, X, z); // the reference's own camera model
			TBAProblem::TObs o;
			o.n_cam = c; o.n_pt = j;
			o.z = z + Eigen::Vector2d(0.5 * RandN(), 0.5 * RandN());
			r_p.obs.push_back(o);
		}
		r_p.points.push_back(X + Eigen::Vector3d(0.02 * RandN(), 0.02 * RandN(), 0.02 * RandN()));
	}
}

template <class CLinearSolverType>
static void Run_BA(const TBAProblem &r_p, bool b_use_schur, std::vector<double> &r_state, size_t n_max_iter)
{
	CBASystemType system;
	CNonlinearSolver_Lambda_LM<CBASystemType, CLinearSolverType> solver(system, TIncrementalSolveSetting(),
		TMarginalsComputationPolicy(), false, CLinearSolverType(), b_use_schur);
	const size_t n_cams = r_p.cams.size();
	for(size_t i = 0; i < n_cams; ++ i)
		system.template r_Get_Vertex<CVertexCam>(i, r_p.cams[i]);
	for(size_t j = 0; j < r_p.points.size(); ++ j)
		system.template r_Get_Vertex<CVertexXYZ>(n_cams + j, r_p.points[j]);
	for(size_t i = 0; i < r_p.obs.size(); ++ i)
		system.r_Add_Edge(CEdgeP2C3D(n_cams + r_p.obs[i].n_pt, r_p.obs[i].n_cam, r_p.obs[i].z,
			Eigen::Matrix2d::Identity(), system));
	solver.Optimize(n_max_iter, 1e-9);
	r_state.clear();
	for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
		Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
		for(int j = 0; j < v.rows(); ++ j)
			r_state.push_back(v(j));
	}
}

static int Compare(const char *p_s_what, size_t a, size_t b, const std::vector<double> &r_ref,
	const std::vector<double> &r_hip, double f_tol)
{
	if(r_ref.size() != r_hip.size() || r_ref.empty())
		return 3;
	double f_max = 0, f_norm = 0;
	for(size_t i = 0; i < r_ref.size(); ++ i) {
		f_max = std::max(f_max, fabs(r_ref[i] - r_hip[i]));
		f_norm = std::max(f_norm, fabs(r_ref[i]));
	}
	printf("%s %lu %lu max_abs_diff %.3e max_abs_state %.3e\n", p_s_what, (unsigned long)a, (unsigned long)b, f_max, f_norm);
	return (f_max <= f_tol * std::max(1.0, f_norm))? 0 : 1;
}

/**
 *	@brief the adapter at scale: a bundle-adjustment-shaped Lambda (n_cams 6 x 6 poses first, n_points 3 x 3 landmarks,
 *		every point seen by n_track consecutive cameras of a sliding window) built directly as a CUberBlockMatrix through
 *		its public interface, diagonally dominant values; CLinearSolver_HIP::Solve_PosDef_Blocky() then walks every block
 *		(Flatten_Values, up to 16 host threads) and solves. Prints what the boundary costs per call; checks the residual.
 */
static int Time_Adapter(size_t n_cams, size_t n_points, size_t n_track)
{
	const size_t nb = n_cams + n_points;
	std::vector<size_t> cumsum(nb);
	for(size_t i = 0, n_sum = 0; i < nb; ++ i)
		cumsum[i] = (n_sum += (i < n_cams)? 6 : 3);
	CUberBlockMatrix lambda(cumsum.begin(), cumsum.end(), cumsum.begin(), cumsum.end());
	const size_t n = cumsum.back();
	size_t n_blocks = 0, n_seed = 12345;
	struct TRand { static double f(size_t &r_s) { r_s = r_s * 6364136223846793005ull + 1442695040888963407ull; return double((r_s >> 33) & 0xffffff) / double(0xffffff) - .5; } };
	for(size_t i = 0; i < n_cams; ++ i) { // camera diagonal blocks: strongly dominant
		double *p_b = lambda.p_FindBlock((i)? cumsum[i - 1] : 0, (i)? cumsum[i - 1] : 0, 6, 6, true, true);
		if(!p_b)
			return 2;
		for(int c = 0; c < 6; ++ c)
			for(int r = 0; r < 6; ++ r)
				p_b[r + 6 * c] = (r == c)? 50.0 * double(n_track) * double(n_points) / double(n_cams) : 0.0;
		++ n_blocks;
	}
	for(size_t j = 0; j < n_points; ++ j) {
		const size_t n_col = cumsum[n_cams + j - 1];
		const size_t n_first = (j * 7919) % (n_cams - n_track + 1); // the track's window
		for(size_t t = 0; t < n_track; ++ t) {
			double *p_b = lambda.p_FindBlock((n_first + t)? cumsum[n_first + t - 1] : 0, n_col, 6, 3, true, true);
			if(!p_b)
				return 2;
			for(int e = 0; e < 18; ++ e)
				p_b[e] = TRand::f(n_seed);
			++ n_blocks;
		}
		double *p_d = lambda.p_FindBlock(n_col, n_col, 3, 3, true, true);
		if(!p_d)
			return 2;
		for(int c = 0; c < 3; ++ c)
			for(int r = 0; r < 3; ++ r)
				p_d[r + 3 * c] = (r == c)? 20.0 * double(n_track) : 0.0;
		++ n_blocks;
	}
	Eigen::VectorXd v_rhs(n), v_x;
	for(size_t i = 0; i < n; ++ i)
		v_rhs(i) = TRand::f(n_seed);
	CLinearSolver_HIP solver;
	double f_best_flat = 1e30, f_best_solve = 1e30;
	for(int n_pass = 0; n_pass < 4; ++ n_pass) {
		v_x = v_rhs;
		if(!solver.Solve_PosDef_Blocky(lambda, v_x)) {
			fprintf(stderr, "error: the factorization failed\n");
			return 2;
		}
		if(n_pass) { // (the first call carries the symbolic analysis)
			f_best_flat = std::min(f_best_flat, solver.f_Last_Flatten_ms());
			f_best_solve = std::min(f_best_solve, solver.f_Last_Solve_ms());
		}
	}
	// residual of the full symmetric system from its upper triangle
	Eigen::VectorXd v_r = -v_rhs;
	for(size_t i = 0, n_col_num = lambda.n_BlockColumn_Num(); i < n_col_num; ++ i) {
		const size_t n_c0 = lambda.n_BlockColumn_Base(i), n_cw = lambda.n_BlockColumn_Column_Num(i);
		for(size_t k = 0, m = lambda.n_BlockColumn_Block_Num(i); k < m; ++ k) {
			const size_t n_row = lambda.n_Block_Row(i, k);
			if(n_row > i)
				continue;
			CUberBlockMatrix::_TyConstMatrixXdRef t_b = ((const CUberBlockMatrix&)lambda).t_Block_AtColumn(i, k);
			const size_t n_r0 = lambda.n_BlockRow_Base(n_row);
			v_r.segment(n_r0, t_b.rows()) += t_b * v_x.segment(n_c0, n_cw);
			if(n_row != i)
				v_r.segment(n_c0, n_cw) += t_b.transpose() * v_x.segment(n_r0, t_b.rows());
		}
	}
	const double f_res = v_r.norm() / v_rhs.norm();
	printf("adapter cams %lu points %lu blocks %lu lambda_mb %.1f flatten_ms %.2f factor_solve_ms %.2f threads %u rel_residual %.3e\n",
		(unsigned long)n_cams, (unsigned long)n_points, (unsigned long)n_blocks, solver.n_Staged_Bytes() * 1e-6, f_best_flat,
		f_best_solve, (unsigned)std::min(16u, std::max(1u, std::thread::hardware_concurrency())), f_res);
	return (f_res < 1e-10)? 0 : 1;
}

int main(int n_arg_num, const char **p_arg_list)
{
	if(n_arg_num > 4 && !strcmp(p_arg_list[1], "adapter"))
		return Time_Adapter(atol(p_arg_list[2]), atol(p_arg_list[3]), atol(p_arg_list[4]));
	if(n_arg_num > 1 && !strcmp(p_arg_list[1], "ba")) {
		size_t n_cams = (n_arg_num > 2)? atol(p_arg_list[2]) : 12;
		size_t n_points = (n_arg_num > 3)? atol(p_arg_list[3]) : 300;
		size_t n_iters = (n_arg_num > 4)? atol(p_arg_list[4]) : 2;
		// BA keeps a gauge freedom (scale) that only the LM damping controls: after many iterations two
		// solvers drift apart along it, so the comparison is made after a few LM iterations
		TBAProblem problem;
		Generate_BA(n_cams, n_points, problem);
		std::vector<double> ref_state, hip_state;
		try {
			Run_BA<CLinearSolver_UberBlock<CBASystemType::_TyHessianMatrixBlockList> >(problem, true, ref_state, n_iters);
			Run_BA<CLinearSolver_HIP>(problem, false, hip_state, n_iters);
		} catch(std::exception &r_exc) {
			fprintf(stderr, "error: %s\n", r_exc.what());
			return 2;
		}
		return Compare("ba cams/points", n_cams, n_points, ref_state, hip_state, 1e-7);
	}
	if(n_arg_num > 5 && !strcmp(p_arg_list[1], "lmdump")) {
		// golden vector of the reference's LEVENBERG-MARQUARDT loop on a generated BA problem (reference
		// linear solver behind the reference's Schur complement, CPU only): the problem, and the vertex
		// states after Optimize(n_iter, 0.01)
		size_t n_cams = atol(p_arg_list[2]), n_points = atol(p_arg_list[3]), n_iters = atol(p_arg_list[4]);
		TBAProblem problem;
		Generate_BA(n_cams, n_points, problem);
		CBASystemType system;
		typedef CLinearSolver_UberBlock<CBASystemType::_TyHessianMatrixBlockList> CLinSolver;
		CNonlinearSolver_Lambda_LM<CBASystemType, CLinSolver> solver(system, TIncrementalSolveSetting(),
			TMarginalsComputationPolicy(), false, CLinSolver(), true);
		for(size_t i = 0; i < n_cams; ++ i)
			system.r_Get_Vertex<CVertexCam>(i, problem.cams[i]);
		for(size_t j = 0; j < problem.points.size(); ++ j)
			system.r_Get_Vertex<CVertexXYZ>(n_cams + j, problem.points[j]);
		for(size_t i = 0; i < problem.obs.size(); ++ i)
			system.r_Add_Edge(CEdgeP2C3D(n_cams + problem.obs[i].n_pt, problem.obs[i].n_cam, problem.obs[i].z,
				Eigen::Matrix2d::Identity(), system));
		solver.Optimize(n_iters, 0.01);
		FILE *p_fw = fopen(p_arg_list[5], "w");
		if(!p_fw)
			return 2;
		fprintf(p_fw, "BALM %lu %lu %lu %lu\n", (unsigned long)n_cams, (unsigned long)problem.points.size(),
			(unsigned long)problem.obs.size(), (unsigned long)n_iters);
		for(size_t i = 0; i < n_cams; ++ i) {
			fprintf(p_fw, "C");
			for(int k = 0; k < 11; ++ k) fprintf(p_fw, " %.17g", problem.cams[i](k));
			fprintf(p_fw, "\n");
		}
		for(size_t j = 0; j < problem.points.size(); ++ j)
			fprintf(p_fw, "P %.17g %.17g %.17g\n", problem.points[j](0), problem.points[j](1), problem.points[j](2));
		for(size_t i = 0; i < problem.obs.size(); ++ i)
			fprintf(p_fw, "O %lu %lu %.17g %.17g\n", (unsigned long)problem.obs[i].n_cam, (unsigned long)problem.obs[i].n_pt,
				problem.obs[i].z(0), problem.obs[i].z(1));
		for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
			Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
			fprintf(p_fw, "F");
			for(int j = 0; j < v.rows(); ++ j) fprintf(p_fw, " %.17g", v(j));
			fprintf(p_fw, "\n");
		}
		fclose(p_fw);
		return 0;
	}
	if(n_arg_num > 4 && !strcmp(p_arg_list[1], "dump3")) {
		// golden vector of the reference's Gauss-Newton loop on a 3D pose graph (poses on rings of a
		// sphere, odometry along each ring + ring-to-ring edges, the shape of sphere2500 at a size the
		// CPU handles in a blink): edges, initial states derived by the reference, states after
		// Optimize(5, 0.01) with the reference's CLinearSolver_UberBlock
		typedef MakeTypelist(CVertexPose3D) TV3;
		typedef MakeTypelist(CEdgePose3D) TE3;
		typedef CFlatSystem<CVertexPose3D, TV3, CEdgePose3D, TE3> CSystem3D;
		typedef Eigen::Matrix<double, 6, 1> V6;
		size_t n_rings = atol(p_arg_list[2]), n_per = atol(p_arg_list[3]);
		std::vector<V6> truth;
		for(size_t r = 0; r < n_rings; ++ r) {
			double phi = -1.0 + 2.0 * (r + 0.5) / n_rings; // latitude-like parameter
			for(size_t k = 0; k < n_per; ++ k) {
				double th = 6.283185307179586 * k / n_per;
				V6 v;
				v << 10 * cos(th) * cos(phi), 10 * sin(th) * cos(phi), 10 * sin(phi), 0.1 * phi, 0.05 * sin(th), th - 3.141592653589793;
				truth.push_back(v);
			}
		}
		struct TE { size_t a, b; V6 z; };
		std::vector<TE> edges3;
		struct L3 { static TE Make(size_t a, size_t b, const std::vector<V6> &t) {
			TE e; e.a = a; e.b = b;
			V6 rel;
			C3DJacobians::Absolute_to_Relative(t[a], t[b], rel);
			for(int k = 0; k < 3; ++ k) { rel(k) += 0.05 * RandN(); rel(3 + k) += 0.01 * RandN(); }
			e.z = rel;
			return e;
		} };
		const size_t n_v = truth.size();
		for(size_t i = 0; i + 1 < n_v; ++ i)
			edges3.push_back(L3::Make(i, i + 1, truth)); // odometry chain through all rings
		for(size_t r = 0; r + 1 < n_rings; ++ r)
			for(size_t k = 0; k < n_per; k += 2)
				edges3.push_back(L3::Make(r * n_per + k, (r + 1) * n_per + k, truth)); // ring-to-ring closures
		for(size_t r = 0; r < n_rings; ++ r)
			edges3.push_back(L3::Make(r * n_per, r * n_per + n_per - 1, truth)); // close every ring
		CSystem3D system;
		CNonlinearSolver_Lambda<CSystem3D, CLinearSolver_UberBlock<CSystem3D::_TyHessianMatrixBlockList> > solver(system);
		Eigen::Matrix<double, 6, 6> information = Eigen::Matrix<double, 6, 6>::Zero();
		for(int k = 0; k < 3; ++ k) { information(k, k) = 400; information(3 + k, 3 + k) = 10000; }
		for(size_t i = 0; i < edges3.size(); ++ i)
			system.r_Add_Edge(CEdgePose3D(edges3[i].a, edges3[i].b, edges3[i].z, information, system));
		std::vector<double> init, fin;
		for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
			Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
			for(int j = 0; j < v.rows(); ++ j) init.push_back(v(j));
		}
		solver.Optimize(5, 0.01);
		for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
			Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
			for(int j = 0; j < v.rows(); ++ j) fin.push_back(v(j));
		}
		FILE *p_fw = fopen(p_arg_list[4], "w");
		if(!p_fw)
			return 2;
		fprintf(p_fw, "SE3GN %lu %lu 5 0.01\n", (unsigned long)(init.size() / 6), (unsigned long)edges3.size());
		for(size_t i = 0; i < edges3.size(); ++ i) {
			fprintf(p_fw, "E %lu %lu", (unsigned long)edges3[i].a, (unsigned long)edges3[i].b);
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", edges3[i].z(k));
			fprintf(p_fw, "\n");
		}
		for(size_t i = 0; i + 5 < init.size(); i += 6) {
			fprintf(p_fw, "I");
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", init[i + k]);
			fprintf(p_fw, "\n");
		}
		for(size_t i = 0; i + 5 < fin.size(); i += 6) {
			fprintf(p_fw, "F");
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", fin[i + k]);
			fprintf(p_fw, "\n");
		}
		fclose(p_fw);
		return 0;
	}
	if(n_arg_num > 3 && !strcmp(p_arg_list[1], "se3dump")) {
		// golden vectors of the reference's SE(3) pose-pose edge geometry (CEdgePose3D,
		// include/slam/SE3_Types.h:264-286): expectation + forward-difference Jacobians of
		// C3DJacobians::Absolute_to_Relative (include/slam/3DSolverBase.h:1331-1371), the edge error, and
		// the vertex (+) (Relative_to_Absolute, :807-850)
		size_t n = atol(p_arg_list[2]);
		FILE *p_fw = fopen(p_arg_list[3], "w");
		if(!p_fw)
			return 2;
		fprintf(p_fw, "SE3GEOM %lu\n", (unsigned long)n);
		typedef Eigen::Matrix<double, 6, 1> V6;
		typedef Eigen::Matrix<double, 6, 6> M6;
		for(size_t i = 0; i < n; ++ i) {
			V6 v1, v2, z, inc, e, err, comp;
			for(int k = 0; k < 3; ++ k) {
				v1(k) = 20 * RandN(); v2(k) = v1(k) + 2 * RandN();
				v1(3 + k) = ((i % 7 == 0)? 1e-5 : 0.9) * RandN();
				v2(3 + k) = v1(3 + k) + ((i % 5 == 0)? 1e-6 : 0.4) * RandN();
				inc(k) = 0.3 * RandN(); inc(3 + k) = 0.1 * RandN();
			}
			M6 H1, H2;
			C3DJacobians::Absolute_to_Relative(v1, v2, e, H1, H2);
			for(int k = 0; k < 3; ++ k) {
				z(k) = e(k) + 0.05 * RandN(); z(3 + k) = e(3 + k) + 0.02 * RandN();
			}
			err.head<3>() = z.head<3>() - e.head<3>();
			Eigen::Quaterniond pQ, dQ;
			C3DJacobians::AxisAngle_to_Quat(z.tail<3>(), pQ);
			C3DJacobians::AxisAngle_to_Quat(e.tail<3>(), dQ);
			Eigen::Vector3d v_aang;
			C3DJacobians::Quat_to_AxisAngle(pQ * dQ.conjugate(), v_aang);
			err.tail<3>() = v_aang;
			C3DJacobians::Relative_to_Absolute(v1, inc, comp);
			fprintf(p_fw, "S");
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", v1(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", v2(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", z(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", e(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", err(k));
			for(int c = 0; c < 6; ++ c) for(int r = 0; r < 6; ++ r) fprintf(p_fw, " %.17g", H1(r, c)); // column-major
			for(int c = 0; c < 6; ++ c) for(int r = 0; r < 6; ++ r) fprintf(p_fw, " %.17g", H2(r, c));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", inc(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", comp(k));
			fprintf(p_fw, "\n");
		}
		fclose(p_fw);
		return 0;
	}
	if(n_arg_num > 3 && !strcmp(p_arg_list[1], "badump")) {
		size_t n = atol(p_arg_list[2]);
		FILE *p_fw = fopen(p_arg_list[3], "w");
		if(!p_fw)
			return 2;
		fprintf(p_fw, "BAGEOM %lu\n", (unsigned long)n);
		for(size_t i = 0; i < n; ++ i) {
			Eigen::Matrix<double, 6, 1> cam, inc, composed;
			Eigen::Matrix<double, 5, 1> intr;
			for(int k = 0; k < 3; ++ k) {
				cam(k) = 2 * RandN();
				cam(3 + k) = ((i % 7 == 0)? 1e-5 : 0.7) * RandN(); // a few near-identity rotations as well
				inc(k) = 0.1 * RandN();
				inc(3 + k) = ((i % 5 == 0)? 1e-7 : 0.05) * RandN();
			}
			intr << 400 + 200 * Rand01(), 400 + 200 * Rand01(), 5 * RandN(), 5 * RandN(), ((i % 3 == 0)? 0.0 : 1e-6 * Rand01());
			Eigen::Vector3d xc(1.5 * RandN(), 1.5 * RandN(), 3 + 5 * Rand01()); // in front of the camera
			Eigen::Matrix3d R = C3DJacobians::t_AxisAngle_to_RotMatrix(cam.tail<3>());
			Eigen::Vector3d X = R.transpose() * (xc - cam.head<3>());
			Eigen::Vector2d uv;
			Eigen::Matrix<double, 2, 6> H1;
			Eigen::Matrix<double, 2, 3> H2;
			CBAJacobians::Project_P2C(cam, intr, X, uv, H1, H2);
			C3DJacobians::Relative_to_Absolute(cam, inc, composed);
			fprintf(p_fw, "S");
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", cam(k));
			for(int k = 0; k < 5; ++ k) fprintf(p_fw, " %.17g", intr(k));
			for(int k = 0; k < 3; ++ k) fprintf(p_fw, " %.17g", X(k));
			for(int k = 0; k < 2; ++ k) fprintf(p_fw, " %.17g", uv(k));
			for(int c = 0; c < 6; ++ c) for(int r = 0; r < 2; ++ r) fprintf(p_fw, " %.17g", H1(r, c)); // column-major
			for(int c = 0; c < 3; ++ c) for(int r = 0; r < 2; ++ r) fprintf(p_fw, " %.17g", H2(r, c));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", inc(k));
			for(int k = 0; k < 6; ++ k) fprintf(p_fw, " %.17g", composed(k));
			fprintf(p_fw, "\n");
		}
		fclose(p_fw);
		return 0;
	}
	if(n_arg_num > 4 && !strcmp(p_arg_list[1], "dump")) {
		size_t n_poses = atol(p_arg_list[2]), n_loops = atol(p_arg_list[3]);
		std::vector<TEdge> edges;
		Generate(n_poses, n_loops, edges);
		std::vector<double> init, fin;
		Run<CLinearSolver_UberBlock<CSystemType::_TyHessianMatrixBlockList> >(edges, fin, 5, 0.01, &init);
		FILE *p_fw = fopen(p_arg_list[4], "w");
		if(!p_fw)
			return 2;
		fprintf(p_fw, "SE2GN %lu %lu 5 0.01\n", (unsigned long)(init.size() / 3), (unsigned long)edges.size());
		for(size_t i = 0; i < edges.size(); ++ i)
			fprintf(p_fw, "E %lu %lu %.17g %.17g %.17g\n", (unsigned long)edges[i].a, (unsigned long)edges[i].b,
				edges[i].z(0), edges[i].z(1), edges[i].z(2));
		for(size_t i = 0; i + 2 < init.size(); i += 3)
			fprintf(p_fw, "I %.17g %.17g %.17g\n", init[i], init[i + 1], init[i + 2]);
		for(size_t i = 0; i + 2 < fin.size(); i += 3)
			fprintf(p_fw, "F %.17g %.17g %.17g\n", fin[i], fin[i + 1], fin[i + 2]);
		fclose(p_fw);
		return 0;
	}
	size_t n_poses = (n_arg_num > 1)? atol(p_arg_list[1]) : 400;
	size_t n_loops = (n_arg_num > 2)? atol(p_arg_list[2]) : 200;
	std::vector<TEdge> edges;
	Generate(n_poses, n_loops, edges);
	std::vector<double> ref_state, hip_state;
	try {
		Run<CLinearSolver_UberBlock<CSystemType::_TyHessianMatrixBlockList> >(edges, ref_state, 5);
		Run<CLinearSolver_HIP>(edges, hip_state, 5);
	} catch(std::exception &r_exc) {
		fprintf(stderr, "error: %s\n", r_exc.what());
		return 2;
	}
	if(ref_state.size() != hip_state.size() || ref_state.empty())
		return 3;
	double f_max = 0, f_norm = 0;
	for(size_t i = 0; i < ref_state.size(); ++ i) {
		f_max = std::max(f_max, fabs(ref_state[i] - hip_state[i]));
		f_norm = std::max(f_norm, fabs(ref_state[i]));
	}
	printf("poses %lu edges %lu max_abs_diff %.3e max_abs_state %.3e\n", (unsigned long)n_poses,
		(unsigned long)edges.size(), f_max, f_norm);
	return (f_max <= 1e-6 * std::max(1.0, f_norm))? 0 : 1;
}
