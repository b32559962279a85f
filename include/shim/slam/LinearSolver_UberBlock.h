/*
 * include/shim/slam/LinearSolver_UberBlock.h -- header-shadowing shim (option i).
 *
 * Put `-I<repo>/include/shim -I<repo>/include` BEFORE the reference's include directory: every
 * `#include "slam/LinearSolver_UberBlock.h"` in the untouched reference sources (e.g.
 * src/slam_simple_example/Main.cpp:24, src/ba_interface_example/BAOptimizer.cpp:26,
 * include/slam/LinearSolver_Schur.h) then resolves to this file, and
 * CLinearSolver_UberBlock<BlockSizes> becomes the MI355X solver. The guard macro is the
 * reference's own (include/slam/LinearSolver_UberBlock.h:14-15), so its header can never be
 * pulled in a second time.
 */
#pragma once
#ifndef __LINEAR_SOLVER_UBERBLOCK_INCLUDED
#define __LINEAR_SOLVER_UBERBLOCK_INCLUDED

#include "spp_adapter.h"

template <class CBlockMatrixTypelist>
class CLinearSolver_UberBlock : public CLinearSolver_HIP {
public:
	typedef CBlockwiseLinearSolverTag _Tag;
	typedef CBlockMatrixTypelist _TyBlockSizes;

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
};

/**
 *	@brief native-solver predicate of the reference (include/slam/LinearSolver_UberBlock.h:429-457);
 *	always false here: the HIP solver does not expose a CUberBlockMatrix factor, so the FastL / DL
 *	solvers take their generic (non-native) code paths
 */
template <class CLinearSolver>
class CIsNativeSolver {
public:
	enum {
		b_result = false
	};
};

#endif // __LINEAR_SOLVER_UBERBLOCK_INCLUDED
