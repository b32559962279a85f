/*
 * oracle/dropin_driver.cpp -- TEST INFRASTRUCTURE. Proves the drop-in boundary end to end:
 * the reference's OWN nonlinear solver (CNonlinearSolver_Lambda::Optimize,
 * include/slam/NonlinearSolver_Lambda.h:476-883) is instantiated twice on the same generated 2D
 * pose graph -- once with the reference's CLinearSolver_UberBlock, once with CLinearSolver_HIP
 * (include/spp_adapter.h -> libspp_hip.so -> MI355X) -- and the optimized vertex states are
 * compared. Built by oracle/Makefile into oracle/_ref/dropin_driver (it contains reference code,
 * so it lives next to libspp_ref.so and never enters git).
 *
 * usage: dropin_driver [n_poses] [n_loop_closures]  ->  prints "max_abs_diff <d> iters <a> <b>"
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#include "slam/LinearSolver_UberBlock.h"
#include "slam/ConfigSolvers.h"
#include "slam/SE2_Types.h"
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
static bool Run(const std::vector<TEdge> &r_edges, std::vector<double> &r_state, size_t n_max_iter)
{
	CSystemType system;
	CNonlinearSolver_Lambda<CSystemType, CLinearSolverType> solver(system);
	Eigen::Matrix3d information;
	information << 1111.11, 0, 0, 0, 1111.11, 0, 0, 0, 10000;
	for(size_t i = 0; i < r_edges.size(); ++ i)
		system.r_Add_Edge(CEdgePose2D(r_edges[i].a, r_edges[i].b, r_edges[i].z, information, system));
	solver.Optimize(n_max_iter, 1e-6);
	r_state.clear();
	for(size_t i = 0, n = system.r_Vertex_Pool().n_Size(); i < n; ++ i) {
		Eigen::VectorXd v = system.r_Vertex_Pool()[i].v_State();
		for(int j = 0; j < v.rows(); ++ j)
			r_state.push_back(v(j));
	}
	return true;
}

int main(int n_arg_num, const char **p_arg_list)
{
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
