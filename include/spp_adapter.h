/*
 * include/spp_adapter.h -- the C++ side of the drop-in boundary: a linear solver class that
 * satisfies SLAM++'s duck-typed solver concept and forwards to the C ABI of libspp_hip.so.
 *
 * Concept implemented (reference include/slam/LinearSolverTags.h:38,54,64-135; model classes
 * include/slam/LinearSolver_UberBlock.h:44-427 and include/slam/LinearSolver_CSparse.h:49-200):
 *   typedef CBlockwiseLinearSolverTag _Tag;
 *   default ctor, copy ctor and operator = that copy configuration only, never state
 *       (LinearSolver_UberBlock.h:74-76,127-130)
 *   void Free_Memory();
 *   bool Solve_PosDef(const CUberBlockMatrix &lambda, Eigen::VectorXd &eta);
 *   void Clear_SymbolicDecomposition();
 *   bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &lambda);
 *   bool Solve_PosDef_Blocky(const CUberBlockMatrix &lambda, Eigen::VectorXd &eta);
 *   bool Factorize_PosDef_Blocky(...)   -- only used by the L / FastL nonlinear solvers
 *       (NonlinearSolver_FastL.h:2131,2388); outside of the accelerated path: supplied by
 *       the shim class include/shim/slam/LinearSolver_UberBlock.h, which delegates to the reference's
 *       own implementation.
 * Error contract: false = not positive definite (the caller prints "Cholesky failed" and stops,
 * NonlinearSolver_Lambda.h:628-664); std::bad_alloc for SPP_E_NOMEM; std::runtime_error otherwise
 * (as LinearSolver_Schur_GPU.cpp:734-797 does for CUDA/CULA errors).
 *
 * Lambda is flattened through the PUBLIC const API of CUberBlockMatrix only
 * (n_BlockColumn_Num, n_BlockColumn_Base, n_BlockColumn_Column_Num, n_BlockColumn_Block_Num,
 * n_Block_Row, t_Block_AtColumn(...).data(); include/slam/BlockMatrix.h:343-485); no pointer into
 * Lambda is retained past a call.
 *
 * When the matrix has the bundle-adjustment structure (two block widths, block-diagonal landmark
 * part) the library eliminates the landmarks itself (Schur complement on the GPU), so this class
 * can be used WITHOUT the reference's CLinearSolver_Schur wrapper / the -us flag.
 *
 * Select it without touching reference sources: see INTEGRATION.md (header shadowing of
 * slam/LinearSolver_CSparse.h or -D__LINEAR_SOLVER_OVERRIDE with the -include shim).
 */
#ifndef SPP_ADAPTER_H
#define SPP_ADAPTER_H

#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <thread>
#include <chrono>

#include "slam/LinearSolverTags.h" // CBlockwiseLinearSolverTag, CUberBlockMatrix, Eigen
#include "spp_hip.h"

class CLinearSolver_HIP {
public:
	typedef CBlockwiseLinearSolverTag _Tag; /**< solver type tag (blockwise: symbolic reuse) */

protected:
	spp_ctx *m_p_ctx;
	int m_n_device, m_n_mode;
	bool m_b_have_symbolic;
	size_t m_n_sym_blocks, m_n_sym_cols;
	std::vector<int64_t> m_col_ptr, m_row_idx, m_blk_off;
	std::vector<int32_t> m_dim;
	double *m_p_vals; /**< flattened block values: page-locked memory owned by the ctx (spp_host_staging) */
	size_t m_n_vals;
	double m_f_flatten_ms, m_f_solve_ms; /**< timing of the last solve (diagnostics) */

public:
	inline CLinearSolver_HIP(int n_device = 0, int n_mode = SPP_MODE_AUTO)
		:m_p_ctx(0), m_n_device(n_device), m_n_mode(n_mode), m_b_have_symbolic(false),
		m_n_sym_blocks(0), m_n_sym_cols(0), m_p_vals(0), m_n_vals(0), m_f_flatten_ms(0), m_f_solve_ms(0)
	{}

	/** copies configuration only, never state (LinearSolver_UberBlock.h:74-76) */
	inline CLinearSolver_HIP(const CLinearSolver_HIP &r_other)
		:m_p_ctx(0), m_n_device(r_other.m_n_device), m_n_mode(r_other.m_n_mode),
		m_b_have_symbolic(false), m_n_sym_blocks(0), m_n_sym_cols(0), m_p_vals(0), m_n_vals(0), m_f_flatten_ms(0), m_f_solve_ms(0)
	{}

	inline ~CLinearSolver_HIP()
	{
		if(m_p_ctx)
			spp_destroy(m_p_ctx);
	}

	inline CLinearSolver_HIP &operator =(const CLinearSolver_HIP &r_other)
	{
		m_n_device = r_other.m_n_device;
		m_n_mode = r_other.m_n_mode;
		return *this;
	}

	inline void Free_Memory()
	{
		if(m_p_ctx)
			spp_free_memory(m_p_ctx);
		m_b_have_symbolic = false;
	}

	inline void Clear_SymbolicDecomposition()
	{
		m_b_have_symbolic = false;
	}

	bool SymbolicDecomposition_Blocky(const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc, std::runtime_error)
	{
		Require_Context();
		Flatten_Structure(r_lambda);
		Check(spp_analyze(m_p_ctx, int64_t(m_dim.size()), &m_col_ptr[0], &m_row_idx[0],
			&m_blk_off[0], &m_dim[0], m_n_mode));
		m_b_have_symbolic = true;
		return true;
	}

	bool Solve_PosDef_Blocky(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		_ASSERTE(r_lambda.b_SymmetricLayout());
		_ASSERTE(size_t(r_eta.rows()) == r_lambda.n_Row_Num());
		if(!m_b_have_symbolic || r_lambda.n_BlockColumn_Num() != m_n_sym_cols ||
		   n_Upper_Block_Num(r_lambda) != m_n_sym_blocks) {
			if(!SymbolicDecomposition_Blocky(r_lambda))
				return false;
		}
		const std::chrono::steady_clock::time_point t_0 = std::chrono::steady_clock::now();
		if(!Flatten_Values(r_lambda)) { // same counts, different structure: the stored symbolic decomposition is stale
			if(!SymbolicDecomposition_Blocky(r_lambda) || !Flatten_Values(r_lambda))
				throw std::runtime_error("libspp_hip adapter: the block structure of lambda changed while it was flattened");
		}
		const std::chrono::steady_clock::time_point t_1 = std::chrono::steady_clock::now();
		int n_result = Check(spp_factor_solve(m_p_ctx, m_p_vals, r_eta.data()));
		m_f_flatten_ms = std::chrono::duration<double>(t_1 - t_0).count() * 1e3;
		m_f_solve_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_1).count() * 1e3;
		return n_result == SPP_OK; // SPP_NOT_POSDEF leaves eta untouched, like the reference
	}

	/** wall clock of the last Solve_PosDef_Blocky(): copying the blocks of lambda into the staging buffer / spp_factor_solve() */
	inline double f_Last_Flatten_ms() const { return m_f_flatten_ms; }
	inline double f_Last_Solve_ms() const { return m_f_solve_ms; }
	inline size_t n_Staged_Bytes() const { return m_n_vals * sizeof(double); }

	bool Solve_PosDef(const CUberBlockMatrix &r_lambda, Eigen::VectorXd &r_eta) // throw(std::bad_alloc, std::runtime_error)
	{
		Clear_SymbolicDecomposition(); // elementwise entry: structure may have changed
		return Solve_PosDef_Blocky(r_lambda, r_eta);
	}


protected:
	inline void Require_Context()
	{
		if(!m_p_ctx) {
			m_p_ctx = spp_create(m_n_device, 0);
			if(!m_p_ctx)
				throw std::runtime_error("spp_create failed: no usable MI355X / HIP device (there is no CPU fallback)");
		}
	}

	inline int Check(int n_code)
	{
		if(n_code >= 0)
			return n_code;
		if(n_code == SPP_E_NOMEM)
			throw std::bad_alloc();
		char p_s_msg[512] = "";
		spp_last_error(m_p_ctx, p_s_msg, sizeof(p_s_msg));
		throw std::runtime_error(std::string("libspp_hip: ") + p_s_msg);
	}

	static size_t n_Upper_Block_Num(const CUberBlockMatrix &r_lambda)
	{
		size_t n_num = 0;
		for(size_t i = 0, n = r_lambda.n_BlockColumn_Num(); i < n; ++ i) {
			for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(i); j < m; ++ j) {
				if(r_lambda.n_Block_Row(i, j) <= i)
					++ n_num;
			}
		}
		return n_num;
	}

	void Flatten_Structure(const CUberBlockMatrix &r_lambda) // throw(std::bad_alloc)
	{
		const size_t n = r_lambda.n_BlockColumn_Num();
		m_dim.resize(n);
		m_col_ptr.resize(n + 1);
		m_row_idx.clear();
		m_blk_off.clear();
		int64_t n_off = 0;
		for(size_t i = 0; i < n; ++ i) {
			m_dim[i] = int32_t(r_lambda.n_BlockColumn_Column_Num(i));
			m_col_ptr[i] = int64_t(m_row_idx.size());
			for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(i); j < m; ++ j) {
				size_t n_row = r_lambda.n_Block_Row(i, j);
				if(n_row > i)
					continue; // only the upper triangle is stored / used
				m_row_idx.push_back(int64_t(n_row));
				m_blk_off.push_back(n_off);
				n_off += int64_t(r_lambda.n_BlockColumn_Column_Num(n_row)) * m_dim[i]; // symmetric layout
			}
		}
		m_col_ptr[n] = int64_t(m_row_idx.size());
		m_n_vals = size_t(n_off);
		m_p_vals = spp_host_staging(m_p_ctx, n_off); // page-locked: spp_factor_solve() then copies at the link's DMA rate
		if(!m_p_vals)
			throw std::bad_alloc();
		m_n_sym_cols = n;
		m_n_sym_blocks = m_row_idx.size();
	}

	/**
	 *	@brief copies the block values; returns false (nothing is trusted then) when the structure of
	 *		r_lambda is not the one of the last symbolic decomposition -- same block counts but other block
	 *		rows or sizes, e.g. one edge replaced by another: the caller re-analyzes and flattens again
	 */
	bool Flatten_Values(const CUberBlockMatrix &r_lambda)
	{
		const size_t n = r_lambda.n_BlockColumn_Num();
		if(n != m_dim.size())
			return false;
		// Large systems (Venice-sized Lambda: 447 MB, 3.4 M blocks) are copied by several host threads, each a contiguous
		// range of block columns: reading a CUberBlockMatrix through its const interface is thread-safe, every thread
		// writes its own range of the staging buffer and the per-block checks stay with the copy.
		size_t n_threads = 1;
		if(m_n_vals * sizeof(double) >= (size_t(32) << 20)) {
			n_threads = std::thread::hardware_concurrency();
			n_threads = (n_threads > 16)? 16 : ((n_threads < 1)? 1 : n_threads);
		}
		if(const char *p_s_env = getenv("SPP_ADAPTER_FLATTEN_THREADS")) // (tests: the threaded copy on small systems too)
			n_threads = size_t((atoi(p_s_env) > 0)? atoi(p_s_env) : 1);
		if(n_threads == 1)
			return Flatten_Columns(r_lambda, 0, n);
		std::vector<char> ok(n_threads, 0);
		std::vector<std::thread> workers;
		for(size_t t = 0; t < n_threads; ++ t) {
			// equal numbers of stored blocks per thread (m_col_ptr is the running block count)
			const int64_t n_b0 = int64_t(m_row_idx.size()) * int64_t(t) / int64_t(n_threads),
				n_b1 = int64_t(m_row_idx.size()) * int64_t(t + 1) / int64_t(n_threads);
			const size_t c0 = (t == 0)? 0 : size_t(std::lower_bound(m_col_ptr.begin(), m_col_ptr.end(), n_b0) - m_col_ptr.begin());
			const size_t c1 = (t + 1 == n_threads)? n : size_t(std::lower_bound(m_col_ptr.begin(), m_col_ptr.end(), n_b1) - m_col_ptr.begin());
			workers.push_back(std::thread([this, &r_lambda, &ok, t, c0, c1]() { ok[t] = Flatten_Columns(r_lambda, c0, (c1 < c0)? c0 : c1)? 1 : 0; }));
		}
		bool b_ok = true;
		for(size_t t = 0; t < n_threads; ++ t) {
			workers[t].join();
			b_ok = b_ok && ok[t] != 0;
		}
		return b_ok;
	}

	/**
	 *	@brief copies the blocks of block columns [n_first, n_end) (a worker of Flatten_Values())
	 */
	bool Flatten_Columns(const CUberBlockMatrix &r_lambda, size_t n_first, size_t n_end) const
	{
		if(n_first >= n_end)
			return true;
		size_t n_blk = size_t(m_col_ptr[n_first]);
		for(size_t i = n_first; i < n_end; ++ i) {
			if(int32_t(r_lambda.n_BlockColumn_Column_Num(i)) != m_dim[i])
				return false;
			for(size_t j = 0, m = r_lambda.n_BlockColumn_Block_Num(i); j < m; ++ j) {
				size_t n_row = r_lambda.n_Block_Row(i, j);
				if(n_row > i)
					continue;
				if(n_blk >= m_row_idx.size() || int64_t(n_blk) >= m_col_ptr[i + 1] || m_row_idx[n_blk] != int64_t(n_row))
					return false;
				CUberBlockMatrix::_TyConstMatrixXdRef t_block = r_lambda.t_Block_AtColumn(i, j);
				if(int32_t(t_block.rows()) != m_dim[n_row] || int32_t(t_block.cols()) != m_dim[i])
					return false;
				memcpy(m_p_vals + size_t(m_blk_off[n_blk]), t_block.data(),
					size_t(t_block.rows()) * size_t(t_block.cols()) * sizeof(double)); // column-major dense block
				++ n_blk;
			}
			if(int64_t(n_blk) != m_col_ptr[i + 1])
				return false;
		}
		return true;
	}
};

#endif // SPP_ADAPTER_H
