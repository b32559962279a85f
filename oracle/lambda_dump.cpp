/*
 * oracle/lambda_dump.cpp -- TEST INFRASTRUCTURE. Pins the ASSEMBLY restatement (oracle/spp_oracle.c:
 * orc_edge_hessians / orc_reduce) and the HIP assembly kernels at the level of Lambda and eta themselves,
 * with numbers produced by the reference's OWN assembly code, not by an end-to-end solve:
 *
 *   - per edge: J0, J1, the error r and the information matrix, as the reference's edge types compute them
 *     (Calculate_Jacobians_Expectation_Error of CEdgePose2D / CEdgePose3D / CEdgeP2C3D), and
 *   - Lambda (upper block triangle) and eta exactly as the reference's nonlinear solver hands them to its
 *     linear solver after CLambdaOps2::Refresh_Lambda (include/slam/NonlinearSolver_Lambda_Base.h:1658-1688,
 *     per-edge products include/slam/BaseTypes_Binary.h:759-848, transposed off-diagonal blocks :783-806,
 *     reduction order _Lambda_Base.h:563-607, unary factor :1903-1924): a RECORDING linear solver -- a class
 *     with the reference's duck-typed solver concept (include/slam/LinearSolverTags.h:38-135) -- copies the
 *     matrix and the right-hand side it is given through the public const API of CUberBlockMatrix and then
 *     delegates to the reference's CLinearSolver_UberBlock.
 *
 * Three small graphs: a 2D pose graph (3 x 3 blocks), a 3D pose graph (6 x 6) and a bundle adjustment problem
 * whose vertex ids interleave cameras and points, so that about half of the camera-point blocks are stored
 * transposed. For BA a second record is taken through CNonlinearSolver_Lambda_LM: Lambda with the
 * Levenberg-Marquardt damping on its diagonal (include/slam/NonlinearSolver_Lambda_LM.h:228-239).
 *
 * A fourth record (ba_robust): the same BA problem with ROBUST edges (Huber weights on Omega, BaseTypes_Binary.h:768-848).
 *
 * usage: lambda_dump se2|se3|ba|ba_robust <out.txt>     (text, %.17g; tools/make_golden_lambda.py -> tests/golden/*_lambda.npz)
 * Built by oracle/Makefile (target `lambda_dump`) into oracle/_ref/; contains reference code, never enters git.
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

#include "slam/LinearSolver_UberBlock.h"
#include "slam/ConfigSolvers.h"
#include "slam/SE2_Types.h"
#include "slam/SE3_Types.h"
#include "slam/BA_Types.h"
#include "slam/NonlinearSolver_Lambda_LM.h"
#include "slam/RobustUtils.h"

static FILE *g_p_out = 0;
static const char *g_p_s_record_name = "LAMBDA";
static int g_n_records_left = 0;

static unsigned long long g_state = 0x9e3779b97f4a7c15ull;
static double Rand01()
{
	g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
	return double(g_state >> 11) / 9007199254740992.0;
}
static double RandN()
{
	double u = Rand01(), v = Rand01();
	return sqrt(-2 * log(u + 1e-300)) * cos(6.283185307179586 * v);
}

static void Record(const CUberBlockMatrix &r_lambda, const Eigen::VectorXd &r_eta)
{
	if(g_n_records_left <= 0)
		return;
	-- g_n_records_left;
	const size_t n = r_lambda.n_BlockColumn_Num();
	size_t n_upper = 0;
	for(size_t i = 0; i < n; ++ i)
		for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(i); j < m; ++ j)
			n_upper += (r_lambda.n_Block_Row(i, j) <= i)? 1 : 0;
	fprintf(g_p_out, "%s %lu %lu %lu\n", g_p_s_record_name, (unsigned long)n, (unsigned long)n_upper, (unsigned long)r_eta.rows());
	fprintf(g_p_out, "DIM");
	for(size_t i = 0; i < n; ++ i)
		fprintf(g_p_out, " %lu", (unsigned long)r_lambda.n_BlockColumn_Column_Num(i));
	fprintf(g_p_out, "\n");
	for(size_t i = 0; i < n; ++ i) {
		for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(i); j < m; ++ j) {
			size_t n_row = r_lambda.n_Block_Row(i, j);
			if(n_row > i)
				continue;
			CUberBlockMatrix::_TyConstMatrixXdRef t_block = r_lambda.t_Block_AtColumn(i, j);
			fprintf(g_p_out, "B %lu %lu", (unsigned long)n_row, (unsigned long)i);
			for(int c = 0; c < t_block.cols(); ++ c)
				for(int r = 0; r < t_block.rows(); ++ r)
					fprintf(g_p_out, " %.17g", t_block(r, c)); // column-major
			fprintf(g_p_out, "\n");
		}
	}
	fprintf(g_p_out, "ETA");
	for(int i = 0; i < r_eta.rows(); ++ i)
		fprintf(g_p_out, " %.17g", r_eta(i));
	fprintf(g_p_out, "\n");
}

/** the reference's solver concept (LinearSolverTags.h:38-135): records, then delegates */
template <class CBlockSizes>
class CLinearSolver_Recorder {
public:
	typedef CBlockwiseLinearSolverTag _Tag;

protected:
	CLinearSolver_UberBlock<CBlockSizes> m_solver;

public:
	CLinearSolver_Recorder() {}
	CLinearSolver_Recorder(const CLinearSolver_Recorder &UNUSED(r_other)) {}
	CLinearSolver_Recorder &operator =(const CLinearSolver_Recorder &UNUSED(r_other)) { return *this; }
	void Free_Memory() { m_solver.Free_Memory(); }
	void Clear_SymbolicDecomposition() { m_solver.Clear_SymbolicDecomposition(); }
	bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &r_lambda) { return m_solver.SymbolicDecomposition_Blocky(r_lambda); }
	bool Solve_PosDef(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta)
	{
		Record(r_lambda, r_eta);
		return m_solver.Solve_PosDef(r_lambda, r_eta);
	}
	bool Solve_PosDef_Blocky(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta)
	{
		Record(r_lambda, r_eta);
		return m_solver.Solve_PosDef_Blocky(r_lambda, r_eta);
	}
};

template <class CEdge, int n_res, int n_d0, int n_d1>
static void Dump_Edge(const CEdge &r_edge)
{
	Eigen::Matrix<double, n_res, n_d0> J0;
	Eigen::Matrix<double, n_res, n_d1> J1;
	Eigen::Matrix<double, n_res, 1> v_expectation, v_error;
	r_edge.Calculate_Jacobians_Expectation_Error(J0, J1, v_expectation, v_error);
	fprintf(g_p_out, "E %lu %lu", (unsigned long)r_edge.n_Vertex_Id(0), (unsigned long)r_edge.n_Vertex_Id(1));
	for(int c = 0; c < n_d0; ++ c) for(int r = 0; r < n_res; ++ r) fprintf(g_p_out, " %.17g", J0(r, c)); // column-major
	for(int c = 0; c < n_d1; ++ c) for(int r = 0; r < n_res; ++ r) fprintf(g_p_out, " %.17g", J1(r, c));
	for(int c = 0; c < n_res; ++ c) for(int r = 0; r < n_res; ++ r) fprintf(g_p_out, " %.17g", r_edge.t_Sigma_Inv()(r, c));
	for(int r = 0; r < n_res; ++ r) fprintf(g_p_out, " %.17g", v_error(r));
	fprintf(g_p_out, "\n");
}

static int Run_SE2()
{
	typedef MakeTypelist(CVertexPose2D) TV;
	typedef MakeTypelist(CEdgePose2D) TE;
	typedef CFlatSystem<CVertexPose2D, TV, CEdgePose2D, TE> CSystem;
	typedef CLinearSolver_Recorder<CSystem::_TyHessianMatrixBlockList> CSolver;
	const size_t n_poses = 60, n_loops = 45;
	std::vector<double> x(n_poses), y(n_poses), th(n_poses);
	int heading = 0;
	x[0] = y[0] = th[0] = 0;
	for(size_t i = 1; i < n_poses; ++ i) {
		if(Rand01() < 0.3)
			heading = (heading + ((Rand01() < 0.5)? 1 : 3)) % 4;
		th[i] = heading * 1.5707963267948966;
		x[i] = x[i - 1] + cos(th[i - 1]);
		y[i] = y[i - 1] + sin(th[i - 1]);
	}
	CSystem system;
	CNonlinearSolver_Lambda<CSystem, CSolver> solver(system);
	Eigen::Matrix3d information;
	information << 1111.11, 12.5, -3.0, 12.5, 900.0, 7.0, -3.0, 7.0, 10000; // full (not diagonal): transposition mistakes show
	std::vector<const CEdgePose2D*> edges;
	struct L { static Eigen::Vector3d z(size_t a, size_t b, const std::vector<double> &x, const std::vector<double> &y,
		const std::vector<double> &th) {
			double c = cos(th[a]), s = sin(th[a]), dx = x[b] - x[a], dy = y[b] - y[a];
			return Eigen::Vector3d(c * dx + s * dy + 0.03 * RandN(), -s * dx + c * dy + 0.03 * RandN(), th[b] - th[a] + 0.01 * RandN());
		} };
	for(size_t i = 0; i + 1 < n_poses; ++ i)
		edges.push_back(&system.r_Add_Edge(CEdgePose2D(i, i + 1, L::z(i, i + 1, x, y, th), information, system)));
	for(size_t k = 0, n_tries = 0; k < n_loops && n_tries < 100000; ++ n_tries) {
		size_t a = size_t(Rand01() * n_poses), b = size_t(Rand01() * n_poses);
		if(a + 1 >= b || b >= n_poses || fabs(x[a] - x[b]) + fabs(y[a] - y[b]) > 4)
			continue;
		edges.push_back(&system.r_Add_Edge(CEdgePose2D(a, b, L::z(a, b, x, y, th), information, system)));
		++ k;
	}
	fprintf(g_p_out, "GRAPH se2 %lu %lu 3 3 3\n", (unsigned long)system.r_Vertex_Pool().n_Size(), (unsigned long)edges.size());
	for(size_t i = 0; i < edges.size(); ++ i)
		Dump_Edge<CEdgePose2D, 3, 3, 3>(*edges[i]);
	g_n_records_left = 1;
	solver.Optimize(2, 1e-9); // the first linear solve is recorded
	return 0;
}

static int Run_SE3()
{
	typedef MakeTypelist(CVertexPose3D) TV;
	typedef MakeTypelist(CEdgePose3D) TE;
	typedef CFlatSystem<CVertexPose3D, TV, CEdgePose3D, TE> CSystem;
	typedef CLinearSolver_Recorder<CSystem::_TyHessianMatrixBlockList> CSolver;
	typedef Eigen::Matrix<double, 6, 1> V6;
	const size_t n_rings = 4, n_per = 9;
	std::vector<V6> truth;
	for(size_t r = 0; r < n_rings; ++ r) {
		double phi = -1.0 + 2.0 * (r + 0.5) / n_rings;
		for(size_t k = 0; k < n_per; ++ k) {
			double t = 6.283185307179586 * k / n_per;
			V6 v;
			v << 10 * cos(t) * cos(phi), 10 * sin(t) * cos(phi), 10 * sin(phi), 0.1 * phi, 0.05 * sin(t), t - 3.141592653589793;
			truth.push_back(v);
		}
	}
	CSystem system;
	CNonlinearSolver_Lambda<CSystem, CSolver> solver(system);
	Eigen::Matrix<double, 6, 6> information = Eigen::Matrix<double, 6, 6>::Zero();
	for(int k = 0; k < 3; ++ k) { information(k, k) = 400 + 10 * k; information(3 + k, 3 + k) = 10000 - 100 * k; }
	information(0, 4) = information(4, 0) = 25; information(1, 2) = information(2, 1) = -8; // full
	std::vector<const CEdgePose3D*> edges;
	struct L { static V6 z(size_t a, size_t b, const std::vector<V6> &t) {
		V6 rel;
		C3DJacobians::Absolute_to_Relative(t[a], t[b], rel);
		for(int k = 0; k < 3; ++ k) { rel(k) += 0.05 * RandN(); rel(3 + k) += 0.01 * RandN(); }
		return rel;
	} };
	const size_t n_v = truth.size();
	for(size_t i = 0; i + 1 < n_v; ++ i)
		edges.push_back(&system.r_Add_Edge(CEdgePose3D(i, i + 1, L::z(i, i + 1, truth), information, system)));
	for(size_t r = 0; r + 1 < n_rings; ++ r)
		for(size_t k = 0; k < n_per; k += 2)
			edges.push_back(&system.r_Add_Edge(CEdgePose3D(r * n_per + k, (r + 1) * n_per + k,
				L::z(r * n_per + k, (r + 1) * n_per + k, truth), information, system)));
	fprintf(g_p_out, "GRAPH se3 %lu %lu 6 6 6\n", (unsigned long)system.r_Vertex_Pool().n_Size(), (unsigned long)edges.size());
	for(size_t i = 0; i < edges.size(); ++ i)
		Dump_Edge<CEdgePose3D, 6, 6, 6>(*edges[i]);
	g_n_records_left = 1;
	solver.Optimize(2, 1e-9);
	return 0;
}

typedef MakeTypelist_Safe((CVertexCam, CVertexXYZ)) TBAVertexTypelist;
typedef MakeTypelist_Safe((CEdgeP2C3D)) TBAEdgeTypelist;
typedef CFlatSystem<CBaseVertex, TBAVertexTypelist, CEdgeP2C3D, TBAEdgeTypelist> CBASystem;

struct TBA {
	std::vector<int> is_cam;                                   // per vertex id
	std::vector<Eigen::Matrix<double, 11, 1> > cam_state;      // per vertex id (cameras only)
	std::vector<Eigen::Vector3d> pt_state;                     // per vertex id (points only)
	struct TObs { size_t n_cam_id, n_pt_id; Eigen::Vector2d z; };
	std::vector<TObs> obs;
};

static void Generate_BA(TBA &r_p)
{
	const size_t n_cams = 7, n_points = 40;
	const size_t n_v = n_cams + n_points;
	// ids interleave: a camera after every fifth point, the remaining cameras at the end
	r_p.is_cam.assign(n_v, 0);
	{
		size_t n_placed = 0;
		for(size_t id = 3; id < n_v && n_placed + 2 < n_cams; id += 6, ++ n_placed)
			r_p.is_cam[id] = 1;
		for(size_t id = n_v; id > 0 && n_placed < n_cams; -- id) {
			if(!r_p.is_cam[id - 1]) { r_p.is_cam[id - 1] = 1; ++ n_placed; }
		}
	}
	r_p.cam_state.resize(n_v);
	r_p.pt_state.resize(n_v);
	std::vector<size_t> cam_ids, pt_ids;
	for(size_t id = 0; id < n_v; ++ id)
		(r_p.is_cam[id]? cam_ids : pt_ids).push_back(id);
	std::vector<Eigen::Matrix<double, 6, 1> > true_cams(n_cams);
	Eigen::Matrix<double, 5, 1> intr;
	intr << 520, 480, 3, -2, 1e-7;
	for(size_t i = 0; i < n_cams; ++ i) {
		double t = 6.283185307179586 * i / n_cams;
		Eigen::Vector3d C(10 * cos(t), 10 * sin(t), 0.5 * sin(3 * t));
		Eigen::Vector3d z = -C.normalized(), x = Eigen::Vector3d(0, 0, 1).cross(z).normalized(), y = z.cross(x);
		Eigen::Matrix3d R;
		R.row(0) = x; R.row(1) = y; R.row(2) = z;
		Eigen::AngleAxisd aa(R);
		true_cams[i].head<3>() = -R * C;
		true_cams[i].tail<3>() = aa.axis() * aa.angle();
		Eigen::Matrix<double, 11, 1> est;
		est.head<6>() = true_cams[i];
		for(int k = 0; k < 3; ++ k) { est(k) += 0.02 * RandN(); est(3 + k) += 0.002 * RandN(); }
		est.tail<5>() = intr;
		r_p.cam_state[cam_ids[i]] = est;
	}
	for(size_t j = 0; j < n_points; ++ j) {
		Eigen::Vector3d X(4 * Rand01() - 2, 4 * Rand01() - 2, 4 * Rand01() - 2);
		size_t k = 2 + size_t(Rand01() * 4), c0 = size_t(Rand01() * n_cams);
		for(size_t q = 0; q < k; ++ q) {
			size_t c = (c0 + q * 2) % n_cams;
			bool b_dup = false;
			for(size_t o = 0; o < r_p.obs.size(); ++ o)
				b_dup = b_dup || (r_p.obs[o].n_pt_id == pt_ids[j] && r_p.obs[o].n_cam_id == cam_ids[c]);
			if(b_dup)
				continue;
			Eigen::Vector2d z;
			CBAJacobians::Project_P2C(true_cams[c], intr, X, z);
			TBA::TObs t_o;
			t_o.n_cam_id = cam_ids[c]; t_o.n_pt_id = pt_ids[j];
			t_o.z = z + Eigen::Vector2d(0.5 * RandN(), 0.5 * RandN());
			r_p.obs.push_back(t_o);
		}
		r_p.pt_state[pt_ids[j]] = X + Eigen::Vector3d(0.02 * RandN(), 0.02 * RandN(), 0.02 * RandN());
	}
}

template <class CSolverType>
static void Fill_BA(CBASystem &r_system, const TBA &r_p, std::vector<const CEdgeP2C3D*> &r_edges)
{
	for(size_t id = 0; id < r_p.is_cam.size(); ++ id) {
		if(r_p.is_cam[id])
			r_system.template r_Get_Vertex<CVertexCam>(id, r_p.cam_state[id]);
		else
			r_system.template r_Get_Vertex<CVertexXYZ>(id, r_p.pt_state[id]);
	}
	Eigen::Matrix2d information;
	information << 1.0, 0.1, 0.1, 0.8;
	for(size_t i = 0; i < r_p.obs.size(); ++ i)
		r_edges.push_back(&r_system.r_Add_Edge(CEdgeP2C3D(r_p.obs[i].n_pt_id, r_p.obs[i].n_cam_id, r_p.obs[i].z, information, r_system)));
}

static int Run_BA()
{
	typedef CLinearSolver_Recorder<CBASystem::_TyHessianMatrixBlockList> CSolver;
	TBA problem;
	Generate_BA(problem);
	{
		CBASystem system;
		CNonlinearSolver_Lambda<CBASystem, CSolver> solver(system);
		std::vector<const CEdgeP2C3D*> edges;
		Fill_BA<CSolver>(system, problem, edges);
		fprintf(g_p_out, "GRAPH ba %lu %lu 2 6 3\n", (unsigned long)system.r_Vertex_Pool().n_Size(), (unsigned long)edges.size());
		for(size_t i = 0; i < edges.size(); ++ i)
			Dump_Edge<CEdgeP2C3D, 2, 6, 3>(*edges[i]);
		g_n_records_left = 1;
		solver.Optimize(2, 1e-9); // Gauss-Newton: Lambda without damping
	}
	{
		CBASystem system;
		CNonlinearSolver_Lambda_LM<CBASystem, CSolver> solver(system, TIncrementalSolveSetting(),
			TMarginalsComputationPolicy(), false, CSolver(), false);
		std::vector<const CEdgeP2C3D*> edges;
		Fill_BA<CSolver>(system, problem, edges);
		g_p_s_record_name = "LAMBDA_LM";
		g_n_records_left = 1;
		solver.Optimize(1, 1e-9); // Levenberg-Marquardt: the damped Lambda of its first linear solve
	}
	return 0;
}

/**
 *	the reference's BA projection edge (include/slam/BA_Types.h:403-560) declared ROBUST the way include/slam/RobustUtils.h:112-125
 *	prescribes -- the CBaseEdge::Robust option of CBaseEdgeImpl plus a CRobustify_* mix-in that supplies f_RobustWeight() --,
 *	here with a Huber kernel on the error norm at a scale of 3/4 pixel so that about half of the test graph's edges get a
 *	weight below one. Every number it produces comes out of the reference's own code: CBAJacobians::Project_P2C, the
 *	kernel, and Calculate_Hessians_v2's robust branch (include/slam/BaseTypes_Binary.h:768-848).
 */
class CEdgeP2C3D_Huber : public CBaseEdgeImpl<CEdgeP2C3D_Huber, MakeTypelist(CVertexCam, CVertexXYZ), 2, 2, CBaseEdge::Robust>,
	public CRobustify_ErrorNorm_Default<CCTFraction<3, 4>, CHuberLossd> {
public:
	typedef CBaseEdgeImpl<CEdgeP2C3D_Huber, MakeTypelist(CVertexCam, CVertexXYZ), 2, 2, CBaseEdge::Robust> _TyBase;
	__GRAPH_TYPES_ALIGN_OPERATOR_NEW

	inline CEdgeP2C3D_Huber()
	{}

	template <class CSystem>
	CEdgeP2C3D_Huber(size_t n_node1, size_t n_node0, const Eigen::Vector2d &v_delta,
		const Eigen::Matrix2d &r_t_inv_sigma, CSystem &r_system)
		:_TyBase(n_node0, n_node1, v_delta, r_t_inv_sigma, CBaseEdge::explicitly_initialized_vertices, r_system)
	{}

	inline void Calculate_Jacobians_Expectation_Error(Eigen::Matrix<double, 2, 6> &r_t_jacobian0,
		Eigen::Matrix<double, 2, 3> &r_t_jacobian1, Eigen::Matrix<double, 2, 1> &r_v_expectation,
		Eigen::Matrix<double, 2, 1> &r_v_error) const
	{
		CBAJacobians::Project_P2C(m_p_vertex0->r_v_State(), m_p_vertex0->v_Intrinsics(),
			m_p_vertex1->r_v_State(), r_v_expectation, r_t_jacobian0, r_t_jacobian1);
		r_v_error = m_v_measurement - r_v_expectation;
	}

	inline double f_Chi_Squared_Error() const
	{
		Eigen::Vector2d v_error;
		CBAJacobians::Project_P2C(m_p_vertex0->r_v_State(), m_p_vertex0->v_Intrinsics(), m_p_vertex1->r_v_State(), v_error);
		v_error -= m_v_measurement;
		return (v_error.transpose() * m_t_sigma_inv).dot(v_error) * f_RobustWeight(v_error);
	}
};

typedef MakeTypelist_Safe((CEdgeP2C3D_Huber)) TBARobustEdgeTypelist;
typedef CFlatSystem<CBaseVertex, TBAVertexTypelist, CEdgeP2C3D_Huber, TBARobustEdgeTypelist> CBARobustSystem;

/** the BA problem of Run_BA() with robust edges: per edge J0 J1 Omega r AND the robust weight, then Lambda / eta */
static int Run_BA_Robust()
{
	typedef CLinearSolver_Recorder<CBARobustSystem::_TyHessianMatrixBlockList> CSolver;
	TBA problem;
	Generate_BA(problem);
	CBARobustSystem system;
	CNonlinearSolver_Lambda<CBARobustSystem, CSolver> solver(system);
	for(size_t id = 0; id < problem.is_cam.size(); ++ id) {
		if(problem.is_cam[id])
			system.r_Get_Vertex<CVertexCam>(id, problem.cam_state[id]);
		else
			system.r_Get_Vertex<CVertexXYZ>(id, problem.pt_state[id]);
	}
	Eigen::Matrix2d information;
	information << 1.0, 0.1, 0.1, 0.8;
	std::vector<const CEdgeP2C3D_Huber*> edges;
	for(size_t i = 0; i < problem.obs.size(); ++ i)
		edges.push_back(&system.r_Add_Edge(CEdgeP2C3D_Huber(problem.obs[i].n_pt_id, problem.obs[i].n_cam_id, problem.obs[i].z, information, system)));
	fprintf(g_p_out, "GRAPH ba_robust %lu %lu 2 6 3\n", (unsigned long)system.r_Vertex_Pool().n_Size(), (unsigned long)edges.size());
	for(size_t i = 0; i < edges.size(); ++ i) {
		Dump_Edge<CEdgeP2C3D_Huber, 2, 6, 3>(*edges[i]);
		Eigen::Matrix<double, 2, 6> J0;
		Eigen::Matrix<double, 2, 3> J1;
		Eigen::Matrix<double, 2, 1> v_expectation, v_error;
		edges[i]->Calculate_Jacobians_Expectation_Error(J0, J1, v_expectation, v_error);
		fprintf(g_p_out, "W %.17g\n", edges[i]->f_RobustWeight(v_error)); // the weight of the edge dumped in the line above
	}
	g_p_s_record_name = "LAMBDA_ROBUST";
	g_n_records_left = 1;
	solver.Optimize(2, 1e-9); // Gauss-Newton: the robustly weighted Lambda of the first linear solve
	return 0;
}

int main(int n_arg_num, const char **p_arg_list)
{
	if(n_arg_num < 3) {
		fprintf(stderr, "usage: lambda_dump se2|se3|ba|ba_robust <out.txt>\n");
		return 1;
	}
	g_p_out = fopen(p_arg_list[2], "w");
	if(!g_p_out)
		return 2;
	int n_result = 1;
	try {
		if(!strcmp(p_arg_list[1], "se2"))
			n_result = Run_SE2();
		else if(!strcmp(p_arg_list[1], "se3"))
			n_result = Run_SE3();
		else if(!strcmp(p_arg_list[1], "ba"))
			n_result = Run_BA();
		else if(!strcmp(p_arg_list[1], "ba_robust"))
			n_result = Run_BA_Robust();
	} catch(std::exception &r_exc) {
		fprintf(stderr, "error: %s\n", r_exc.what());
		n_result = 3;
	}
	fclose(g_p_out);
	return n_result;
}
