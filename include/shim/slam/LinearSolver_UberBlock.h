/*
 * include/shim/slam/LinearSolver_UberBlock.h -- header-shadowing shim (option i).
 *
 * Put `-I<repo>/include/shim -I<repo>/include` BEFORE the reference's include directory: every
 * `#include "slam/LinearSolver_UberBlock.h"` in the untouched reference sources (e.g.
 * src/slam_simple_example/Main.cpp:24, src/ba_interface_example/BAOptimizer.cpp:26,
 * include/slam/LinearSolver_Schur.h) then resolves to this file, and
 * CLinearSolver_UberBlock<BlockSizes> becomes the MI355X solver. slam_app selects it with
 * -D__LINEAR_SOLVER_OVERRIDE=3 (include/slam_app/Config.h:90-109, Main.h:1116-1121), the examples name
 * the type directly.
 */
#pragma once
#ifndef SPP_SHIM_LINEAR_SOLVER_UBERBLOCK_INCLUDED
#define SPP_SHIM_LINEAR_SOLVER_UBERBLOCK_INCLUDED

/*
 * The reference's own class is still needed for ONE method that is not on the accelerated path:
 * Factorize_PosDef_Blocky (include/slam/LinearSolver_UberBlock.h:216-258), which the L / FastL
 * nonlinear solvers call on their second linear solver (NonlinearSolver_FastL.h:2131,2388,
 * NonlinearSolver_L.h:1663) and which slam_app instantiates for every linear solver type
 * (ConfigSolvers.h:272-279). Its header is pulled in untouched -- #include_next continues the search
 * behind this directory -- with the class renamed by the preprocessor for the duration of that
 * include; the native-solver predicate CIsNativeSolver (:429-457) comes with it and is true for the
 * renamed reference class only, so FastL / DL take their generic code paths with the HIP solver.
 */
#define CLinearSolver_UberBlock CLinearSolver_UberBlock_Reference
#include_next "slam/LinearSolver_UberBlock.h"
#undef CLinearSolver_UberBlock

#include "spp_adapter.h"

template <class CBlockMatrixTypelist>
class CLinearSolver_UberBlock : public CLinearSolver_HIP {
public:
	typedef CBlockwiseLinearSolverTag _Tag;
	typedef CBlockMatrixTypelist _TyBlockSizes;

protected:
	CLinearSolver_UberBlock_Reference<CBlockMatrixTypelist> m_reference_factorizer; /**< only for Factorize_PosDef_Blocky() */

public:
	inline CLinearSolver_UberBlock()
	{}

	inline CLinearSolver_UberBlock(const CLinearSolver_UberBlock &r_other)
		:CLinearSolver_HIP(r_other)
	{}

	inline CLinearSolver_UberBlock &operator =(const CLinearSolver_UberBlock &r_other)
	{
		CLinearSolver_HIP::operator =(r_other);
		return *this;
	}

	inline void Free_Memory()
	{
		CLinearSolver_HIP::Free_Memory();
		m_reference_factorizer.Free_Memory();
	}

	/**
	 *	@brief factor of a pre-ordered block matrix as a block matrix: outside of the accelerated path
	 *		(the HIP solver keeps its factor in frontal form in HBM), delegated to the reference's own
	 *		implementation (include/slam/LinearSolver_UberBlock.h:216-258), same arguments and result
	 */
	inline bool Factorize_PosDef_Blocky(CUberBlockMatrix &r_factor, const CUberBlockMatrix &r_lambda,
		std::vector<size_t> &r_workspace, size_t n_dest_row_id = 0,
		size_t n_dest_column_id = 0, bool b_upper_factor = true) // throw(std::bad_alloc)
	{
		return m_reference_factorizer.Factorize_PosDef_Blocky(r_factor, r_lambda, r_workspace,
			n_dest_row_id, n_dest_column_id, b_upper_factor);
	}
};

#endif // SPP_SHIM_LINEAR_SOLVER_UBERBLOCK_INCLUDED
