/*
 * include/spp_hip.h -- C ABI of libspp_hip.so, the MI355X (gfx950) block-sparse Lambda solver
 * that drops in behind SLAM++'s duck-typed CLinearSolver_* concept.
 *
 * The reference has NO C ABI for this path: a linear solver is a C++ template concept
 * (reference: include/slam/LinearSolverTags.h:38,54,64-135; the native model is
 * include/slam/LinearSolver_UberBlock.h:44-427, the Schur wrapper include/slam/LinearSolver_Schur.h:1423-2392).
 * include/spp_adapter.h implements that concept on top of these entry points; INTEGRATION.md
 * shows how the reference is recompiled against it without source edits.
 *
 * Conventions
 *   - every function returns SPP_OK (0), SPP_NOT_POSDEF (1: the adapter maps it to `return false`,
 *     reference: BlockMatrix.cpp:9765-9771 / NonlinearSolver_Lambda.h:628-664), or a negative
 *     SPP_E_* code (the adapter throws std::runtime_error / std::bad_alloc, reference:
 *     LinearSolver_Schur_GPU.cpp:734-797); spp_last_error() returns the message.
 *   - plain pointers and sizes only; int64 block indices / offsets, int32 block dims, fp64 values.
 *   - pointers named h_* are host memory, d_* are device (HBM) memory of the ctx's device.
 *   - one ctx = one device + one stream, used from one thread (reference threading contract:
 *     LinearSolver_Schur.h:1187 binds one context per solver instance).
 *   - there is NO CPU fallback: every compute entry point fails with SPP_E_NO_DEVICE if HIP
 *     cannot run on the selected device.
 *
 * Matrix layout (the flattening of CUberBlockMatrix, reference BlockMatrixBase.h:380-503):
 *   upper-triangular block-CSC; nb block columns; dim[nb] widths; col_ptr[nb+1]; row_idx[nnzb]
 *   ascending per column with the diagonal block last; blk_off[nnzb] = offset in doubles of block
 *   p inside `vals`; a block is dim[row] x dim[col], column-major (the element order
 *   t_Block_AtColumn(...).data() yields, BlockMatrix.h:343-485).
 */
#ifndef SPP_HIP_H
#define SPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPP_OK            0
#define SPP_NOT_POSDEF    1
#define SPP_E_BADARG     -1
#define SPP_E_NOMEM      -2
#define SPP_E_HIP        -3
#define SPP_E_NO_DEVICE  -4
#define SPP_E_STATE      -5   /* call order violated (e.g. solve before analyze) */
#define SPP_E_UNSUPPORTED -6

/* spp_analyze modes */
#define SPP_MODE_AUTO     0   /* Schur if two block widths with a block-diagonal landmark part, else sparse */
#define SPP_MODE_SPARSE   1   /* ordering + supernodal block Cholesky on the whole Lambda */
#define SPP_MODE_SCHUR    2   /* guided Schur complement (landmarks = smaller width) + dense reduced solve */
#define SPP_MODE_SCHUR_SPARSE 3 /* guided Schur complement, the reduced camera system kept sparse (block-CSC) and
                                   solved by the supernodal path: CLinearSolver_Schur with a sparse inner solver,
                                   include/slam/LinearSolver_Schur.h:1844-1853. AUTO picks it beyond 16384 reduced
                                   scalars (BASELINE config 5: 10k cameras). Shards over landmarks like SPP_MODE_SCHUR: every
                                   rank holds the union block structure of S, the all-reduced buffer is the block value
                                   array followed by the reduced rhs */

/* spp_create flags */
#define SPP_FLAG_PROFILE  1   /* record hipEvents around the phases of each solve */
#define SPP_FLAG_SCHUR_PARTIAL 2 /* multi-GPU: spp_schur_form() leaves the partial S / rhs for an external all-reduce */

typedef struct spp_ctx spp_ctx;

/* ---- lifetime ---------------------------------------------------------------------------------
 * replaces: CLinearSolver_UberBlock ctor / dtor / Free_Memory (LinearSolver_UberBlock.h:69-121) */
spp_ctx *spp_create(int device, int flags);
void spp_destroy(spp_ctx *ctx);
/* drops factor/workspaces but keeps the ctx usable (Free_Memory, LinearSolver_UberBlock.h:88-121) */
int spp_free_memory(spp_ctx *ctx);
int spp_last_error(const spp_ctx *ctx, char *buf, size_t buf_size);
/* Page-locked host buffer of at least n_doubles doubles, owned by the ctx (grows, freed by spp_destroy), or NULL
 * when it cannot be had. The adapter flattens Lambda's blocks straight into it (the role of the m_vals vector) so
 * that the host-pointer entry spp_factor_solve() copies to the device at the DMA rate of the link instead of through
 * the runtime's pageable bounce buffers: the reference's own workspace reuse, LinearSolver_UberBlock.h:332-348. */
double *spp_host_staging(spp_ctx *ctx, int64_t n_doubles);
/* run all work of this ctx on an externally owned hipStream_t (e.g. torch's current stream) */
int spp_set_stream(spp_ctx *ctx, void *hip_stream);
int spp_synchronize(spp_ctx *ctx);

/* ---- symbolic -----------------------------------------------------------------------------------
 * replaces: SymbolicDecomposition_Blocky (LinearSolver_UberBlock.h:272-296: block AMD ordering via
 * OrderingMagic.cpp:701; LinearSolver_Schur.h:1566-1606: guided ordering) plus the structure-only
 * work the reference redoes inside every Solve_PosDef_Blocky: Permute_UpperTriangular_To
 * (BlockMatrix.cpp:8183), Build_EliminationTree (:9403), ereach (:9453), SliceTo/TransposeTo
 * (LinearSolver_Schur.h:1699-1709). Call again whenever the block structure changes
 * (Clear_SymbolicDecomposition, LinearSolver_UberBlock.h:260-264). */
int spp_analyze(spp_ctx *ctx, int64_t nb, const int64_t *h_col_ptr, const int64_t *h_row_idx,
	const int64_t *h_blk_off, const int32_t *h_dim, int mode);

/* multi-GPU landmark sharding: this rank keeps only landmarks p with
 * shard_of_landmark == rank; cameras and A are replicated. Call before spp_analyze. */
int spp_set_shard(spp_ctx *ctx, int rank, int world_size);

/* facts about the analyzed system; unknown keys give SPP_E_BADARG */
#define SPP_MODE_SCHUR_MIS 4  /* Schur complement over a maximal independent set of a ONE-width graph (3 or 6: pose
                                 graphs): the general ordering of CSchurOrdering (src/slam/LinearSolver_Schur.cpp:690-769,
                                 1235-1340); reduced system sparse, supernodal solve. Never chosen by AUTO */

#define SPP_INFO_MODE           0  /* SPP_MODE_SPARSE, SPP_MODE_SCHUR or SPP_MODE_SCHUR_SPARSE actually chosen */
#define SPP_INFO_N              1  /* scalar dimension */
#define SPP_INFO_NNZB           2  /* stored blocks of Lambda (upper incl. diagonal) */
#define SPP_INFO_NVALS          3  /* doubles in vals */
#define SPP_INFO_FACTOR_NNZ     4  /* scalar nonzeros of the factor (sparse) or n_reduced^2 storage (Schur dense) */
#define SPP_INFO_FACTOR_FLOPS   5  /* flops of the numeric factorization */
#define SPP_INFO_N_REDUCED      6  /* Schur: dimension of the reduced camera system */
#define SPP_INFO_N_POSES        7
#define SPP_INFO_N_LANDMARKS    8  /* landmarks owned by this shard */
#define SPP_INFO_SCHUR_PAIRS    9  /* sum_p k_p (k_p+1)/2 block products of the S accumulation */
#define SPP_INFO_N_OBS          10 /* pose-landmark blocks owned by this shard */
#define SPP_INFO_SOLVE_BYTES    11 /* algorithmic HBM bytes of one numeric solve */
#define SPP_INFO_N_SUPERNODES   12
#define SPP_INFO_N_LEVELS       13
#define SPP_INFO_S_LD           14 /* leading dimension of the dense S buffer (padded); 0 when S is sparse */
#define SPP_INFO_S_NNZB         15 /* Schur: stored blocks of the reduced camera system (upper incl. diagonal) */
#define SPP_INFO_DENSE_STREAMED 16 /* tile rows (of 128) the last dense factorization handed to the streamed launch (spp_dense_tail.h); 0: none */
int spp_get_info(const spp_ctx *ctx, int what, int64_t *out);
/* elimination order chosen by the analysis: order[k] = source block column eliminated k-th */
int spp_get_ordering(const spp_ctx *ctx, int64_t *h_order);

/* Fill-reducing block ordering of an upper block pattern, host only (no context, no GPU): the job of
 * CMatrixOrdering::p_BlockOrdering (reference src/slam/OrderingMagic.cpp:701, live path :900-1031:
 * A + A^T block pattern -> AMD). order[k] = block column eliminated k-th (the reference returns the
 * inverse of this). SPP_ORDER_AMD: approximate minimum degree (quotient graph); SPP_ORDER_ND: nested
 * dissection on BFS level structures with minimum-degree leaves (logarithmic tree height on
 * chain-like graphs). The analysis picks between the two itself (ordering). */
#define SPP_ORDER_AMD 0
#define SPP_ORDER_ND  1
int spp_block_ordering(int64_t nb, const int64_t *col_ptr, const int64_t *row_idx, int method, int64_t *h_order);

/* The symbolic Schur plan of an upper block pattern, host only (no context, no GPU): what the reference recomputes
 * structurally in every CLinearSolver_Schur::Solve_PosDef_Blocky call (guided ordering LinearSolver_Schur.cpp:771-838,
 * the slices LinearSolver_Schur.h:1699-1709, the symbolic part of MultiplyToWith BlockMatrixFBS.inl:1147-1304) as
 * observation lists, the block pattern of S and its per-block lists of block products. Runs on up to 16 host threads
 * (SPP_PLAN_THREADS); the result does not depend on their number. out[0..7] = poses, landmarks of this shard,
 * observations, block products, blocks of S, work items, split blocks, a 64-bit checksum of the lists; *seconds = wall
 * clock of the plan. For tests and for timing the analysis phase without a device. */
int spp_schur_plan_host(int64_t nb, const int32_t *dim, const int64_t *col_ptr, const int64_t *row_idx, int shard_rank,
	int shard_world, int sparse_S, int64_t *out, double *seconds);

/* ---- numeric: host-pointer entry points (what the header adapter calls) ----------------------------
 * replaces: Solve_PosDef_Blocky (LinearSolver_UberBlock.h:312-426; LinearSolver_Schur.h:1623-1935)
 * and Solve_PosDef (LinearSolver_UberBlock.h:143-258). h_vals holds the blocks at the blk_off
 * given to spp_analyze; h_rhs (length n) is overwritten with the solution. */
int spp_factor_solve(spp_ctx *ctx, const double *h_vals, double *h_rhs_inout);

/* ---- numeric: device-resident entry points (Lambda lives in HBM across GN iterations) ------------
 * d_vals: nvals doubles in the layout given to spp_analyze; d_rhs: n doubles, in/out (contents unspecified after
 * SPP_NOT_POSDEF: the status is fetched once, after the solves have run). */
int spp_factor_solve_device(spp_ctx *ctx, const double *d_vals, double *d_rhs_inout);

/* split form for multi-GPU: (1) each rank forms its partial Schur complement and
 * reduced rhs into d_S_rhs = [S (ld x ld, column-major, upper blocks) | rhs (ld)] ; (2) the caller
 * all-reduces that buffer (RCCL); (3) every rank factors S, solves and back-substitutes its own
 * landmarks. With world_size == 1 step (2) is a no-op. rank 0 alone adds A and the pose rhs. */
int spp_schur_buffer_size(const spp_ctx *ctx, int64_t *n_doubles);
int spp_schur_form(spp_ctx *ctx, const double *d_vals, const double *d_rhs, double *d_S_rhs);
int spp_schur_finish(spp_ctx *ctx, const double *d_vals, double *d_S_rhs, double *d_rhs_inout);
/* Only the upper block-trapezoid of S carries data: pack it (128-column panels, rows 0 .. end of the
 * panel's diagonal block; 128^2 nblk (nblk + 1) / 2 doubles, about half of the square buffer) before
 * the all-reduce and unpack afterwards -- halves the bytes that cross xGMI. */
int spp_schur_packed_size(const spp_ctx *ctx, int64_t *n_doubles);
int spp_schur_pack(spp_ctx *ctx, const double *d_S_rhs, double *d_packed);
int spp_schur_unpack(spp_ctx *ctx, const double *d_packed, double *d_S_rhs);

/* ---- Lambda / eta assembly ------------------------------------------------------------------------
 * replaces: CLambdaOps2::AddEntriesInSparseSystem + Alloc_HessianBlocks_v2 (symbolic;
 * NonlinearSolver_Lambda_Base.h:1852-1931, BaseTypes_Binary.h:525-660) and Refresh_Lambda =
 * Calculate_Hessians_v2 over all edges + ReduceAll (numeric; _Lambda_Base.h:1658-1688,
 * BaseTypes_Binary.h:759-848, _Lambda_Base.h:563-607,152-197).
 * One homogeneous binary-edge group: residual dimension rd, vertex 0 width d0, vertex 1 width d1.
 * J0: ne x (rd x d0) col-major, J1: ne x (rd x d1), Omega: ne x (rd x rd), r: ne x rd.
 * The unary factor (identity, FlatSystem.h:441,467) is added to the diagonal block of vertex
 * `unary_vertex` (pass -1 for none). The reference's default build puts it on vertex 0 whatever its type
 * (__AUTO_UNARY_FACTOR_ON_VERTEX_ZERO, FlatSystem.h:331-337, _Lambda_Base.h:1903-1924; without that macro: on the
 * first vertex of the first edge). `damping` is added to every diagonal entry of Lambda
 * (Levenberg-Marquardt, NonlinearSolver_Lambda_LM.h:228-239; 0 for Gauss-Newton). */
int spp_assemble_analyze(spp_ctx *ctx, int64_t nv, const int32_t *h_dim,
	int64_t ne, const int64_t *h_v0, const int64_t *h_v1, int d0, int d1, int rd,
	int64_t unary_vertex);
/* structure produced by spp_assemble_analyze (sizes via spp_get_info NNZB / NVALS) */
int spp_assemble_get_structure(const spp_ctx *ctx, int64_t *h_col_ptr, int64_t *h_row_idx,
	int64_t *h_blk_off);
int spp_assemble_device(spp_ctx *ctx, const double *d_J0, const double *d_J1,
	const double *d_Omega, const double *d_r, double damping, double *d_vals_out, double *d_eta_out);
/* Robust edges (the reference's b_is_robust_edge branch of Calculate_Hessians_v2, include/slam/BaseTypes_Binary.h:768-848;
 * kernels and mix-ins include/slam/RobustUtils.h): d_w holds ONE weight per edge -- the value of the edge's
 * f_RobustWeight(r), evaluated by the caller -- and every following spp_assemble_device applies it exactly where the
 * reference does: H01 and H00 carry w once (T = J0^T Omega w), the first vertex's right-hand side w TWICE (T r w),
 * H11 and the second vertex's right-hand side once. The array is NOT copied (it must stay valid); NULL restores plain
 * edges; spp_assemble_analyze resets it. */
int spp_assemble_set_edge_weights(spp_ctx *ctx, const double *d_w);
/* the weights themselves, on the device: w_e = kernel(||r_e|| / scale) -- CRobustify_ErrorNorm_Default::f_RobustWeight
 * (include/slam/RobustUtils.h:396-400) with kind 0 = Huber, w = 1 for x <= param, param / x beyond (CHuberLoss::operator (),
 * include/geometry/RobustLoss.h:100-104; the reference's default param is 1.345). Asynchronous on the ctx stream. */
int spp_edge_robust_weights_device(spp_ctx *ctx, int64_t n_edges, int rd, int kind, double scale, double param,
	const double *d_r, double *d_w_out);

/* ---- device memory helpers for hosts without a HIP runtime of their own ---------------------------- */
/* ---- on-device geometry of 2D pose graphs (CEdgePose2D) --------------------------
 * spp_se2_linearize_device: per edge (pose v0 -> pose v1, measurement z = dx dy dtheta in the frame of
 * v0) the Jacobians of the expectation and the error r = z - h(x), exactly the quantities of
 * C2DJacobians::Absolute_to_Relative (include/slam/2DSolverBase.h:373-418) and CEdgePose2D (angle error
 * wrapped by f_ClampAngularError_2Pi, :90-94), written in the layout spp_assemble_device reads
 * (J0, J1: ne x 3x3 column-major, r: ne x 3). All pointers are device pointers; v0 / v1 are int32.
 * spp_se2_update_device: ||dx||^2 -> *h_dx_norm2 (deterministic two-stage sum) and, if `apply`,
 * x <- x (+) dx with the angle clamped (CVertexPose2D::Operator_Plus, include/slam/SE2_Types.h:70-74).
 * It synchronizes the stream: the stopping test of the Gauss-Newton loop needs the norm on the host
 * (NonlinearSolver_Lambda.h:638-650). */
int spp_se2_linearize_device(spp_ctx *ctx, int64_t n_edges, const int32_t *d_v0, const int32_t *d_v1,
	const double *d_poses, const double *d_measurements, double *d_J0, double *d_J1, double *d_r);
int spp_se2_update_device(spp_ctx *ctx, int64_t n_vertices, double *d_poses, const double *d_dx, int apply,
	double *h_dx_norm2);

/* ---- on-device geometry of 3D pose graphs (CEdgePose3D) ---------------------------
 * Poses: 6 doubles [t | axis-angle]. Expectation = C3DJacobians::Absolute_to_Relative(v0, v1), error
 * [z_t - e_t ; log(R(z_r) R(e_r)^T)] (include/slam/SE3_Types.h:264-286), Jacobians w.r.t. the increments of
 * Relative_to_Absolute (3DSolverBase.h:807-850) -- analytic here, forward differences with delta = 1e-9 in
 * the reference (:1331-1371). J0, J1: ne x (6x6) column-major, r: ne x 6 (the (6,6,6) group of
 * spp_assemble_device). spp_se3_update_device: ||dx||^2 and x <- x (+) dx (CVertexPose3D::Operator_Plus). */
int spp_se3_linearize_device(spp_ctx *ctx, int64_t n_edges, const int32_t *d_v0, const int32_t *d_v1,
	const double *d_poses, const double *d_measurements, double *d_J0, double *d_J1, double *d_r);
int spp_se3_update_device(spp_ctx *ctx, int64_t n_vertices, double *d_poses, const double *d_dx, int apply,
	double *h_dx_norm2);

/* ---- on-device geometry of bundle adjustment (CEdgeP2C3D) -------------------------
 * Cameras: 6 doubles each [t | axis-angle], world -> camera, and 5 constant intrinsics each (fx fy cx cy k:
 * CVertexCam, include/slam/BA_Types.h); points: XYZ. spp_ba_linearize_device evaluates per observation the
 * projection of CBAJacobians::Project_P2C (include/slam/BASolverBase.h:260-325), r = z - uv, and its
 * Jacobians w.r.t. the camera increment of C3DJacobians::Relative_to_Absolute (t' = t + R dt,
 * R' = R exp(dr), include/slam/3DSolverBase.h:807-850) and w.r.t. the point -- analytically, where the
 * reference takes forward differences with delta = 1e-9 (BASolverBase.h:559-620): agreement to ~1e-7
 * relative, the noise of the difference quotients. Output layout = input of spp_assemble_device for the
 * (6,3,2) edge group: J0 no x (2x6) column-major, J1 no x (2x3), r no x 2. cam_of / pt_of: int32 indices
 * into the camera / point arrays.
 * spp_ba_update_device: ||dx||^2 over the n_dx entries of dx -> *h_dx_norm2, and if `apply` camera i
 * <- camera i (+) dx[cam_dxoff[i] .. +6) (the composition above), point j += dx[pt_dxoff[j] .. +3)
 * (CVertexCam / CVertexXYZ::Operator_Plus). Synchronizes the stream. */
int spp_ba_linearize_device(spp_ctx *ctx, int64_t n_obs, const int32_t *d_cam_of, const int32_t *d_pt_of,
	const double *d_cams, const double *d_intrinsics, const double *d_points, const double *d_measurements,
	double *d_J0, double *d_J1, double *d_r);
int spp_ba_update_device(spp_ctx *ctx, int64_t n_cams, double *d_cams, const int64_t *d_cam_dxoff,
	int64_t n_points, double *d_points, const int64_t *d_pt_dxoff, const double *d_dx, int64_t n_dx, int apply,
	double *h_dx_norm2);

/* ---- scalars of the Levenberg-Marquardt control (include/slam/NonlinearSolver_Lambda_LM.h) ----------
 * chi2 = sum_e r_e^T Omega_e r_e (f_Error, :1078-1095); the largest diagonal entry of any vertex Hessian
 * J_i^T Omega J_i over all edges (f_InitialDamping multiplies it by tau = 1e-3, :151-199); the denominator
 * dx . (alpha dx + eta) of the gain ratio (Aftermath, :204-222). Device inputs, host outputs, deterministic
 * reductions; each call synchronizes the stream. rd / (d0, d1): the edge group of spp_assemble_analyze. */
int spp_edge_chi2_device(spp_ctx *ctx, int64_t n_edges, int rd, const double *d_r, const double *d_Omega, double *h_chi2);
int spp_edge_hessian_maxdiag_device(spp_ctx *ctx, int64_t n_edges, int rd, int d0, int d1, const double *d_J0,
	const double *d_J1, const double *d_Omega, double *h_max);
int spp_lm_gain_denominator_device(spp_ctx *ctx, int64_t n, const double *d_dx, const double *d_eta, double alpha,
	double *h_out);

int spp_device_malloc(spp_ctx *ctx, size_t bytes, void **d_ptr);
int spp_device_free(spp_ctx *ctx, void *d_ptr);
int spp_memcpy_h2d(spp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int spp_memcpy_d2h(spp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* asynchronous on the ctx stream */
int spp_memcpy_d2d(spp_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);

/* switch SPP_FLAG_PROFILE on / off for the following solves (the hipEvents around phases and around the
 * dominant kernel cost microseconds per solve; a benchmark times with profiling off and reads the
 * breakdown from a separate profiled pass) */
int spp_set_profiling(spp_ctx *ctx, int on);

/* ---- profiling (SPP_FLAG_PROFILE) ------------------------------------------------------------------
 * phase names follow the reference's __SCHUR_PROFILING / Dump() vocabulary
 * (LinearSolver_Schur.h:1889-1912, NonlinearSolver_Lambda.h:250-276). */
#define SPP_PHASE_PERMUTE   0
#define SPP_PHASE_SCHUR_INV 1  /* C^-1 and W = -U C^-1 */
#define SPP_PHASE_SCHUR_GEMM 2 /* S = A + W U^T */
#define SPP_PHASE_SCHUR_RHS 3
#define SPP_PHASE_FACTOR    4  /* numeric Cholesky (dense or sparse) */
#define SPP_PHASE_TRISOLVE  5
#define SPP_PHASE_BACKSUBST 6  /* landmark back-substitution */
#define SPP_PHASE_ASSEMBLE  7
#define SPP_PHASE_TOTAL     8
#define SPP_N_PHASES        9
/* milliseconds of each phase of the LAST solve / assemble call (hipEvent elapsed on the ctx stream) */
int spp_get_phase_ms(spp_ctx *ctx, double *ms_out /* [SPP_N_PHASES] */);
/* hipEvent-timed average duration (ms) of the dominant kernel family of the last solve:
 * the MFMA trailing-update GEMM launches of the dense factor; n_launches/flops are totals */
int spp_get_dominant_kernel(spp_ctx *ctx, double *ms_total, int64_t *n_launches, double *flops);

/* ---- micro-benchmarks used by bench.py to report measured peaks beside the spec peaks ------------ */
int spp_microbench_copy(spp_ctx *ctx, size_t bytes, int iters, double *gb_per_s);
int spp_microbench_mfma_f64(spp_ctx *ctx, int iters, double *tflops);
/* read-negate-write of an n x n fp64 matrix (n % 128 == 0) in the trailing-update kernel's C-tile lane
   pattern; a known byte count (2 * 8 * n * n per launch) to calibrate the PMC traffic counters with */
int spp_microbench_ctile(spp_ctx *ctx, int n, int iters, double *gb_per_s);
/* the bulk trailing update of ONE dense factorization step, stand-alone: an m x (m + 1) trailing matrix (upper
 * tiles) receives a rank-128 update `iters` times back to back; *ms_per_launch = hipEvent time per launch.
 * Useful flops per launch = 128 m (m + 1) + 256 m (what bench.py's roofline counts for the same launch). */
int spp_microbench_update(spp_ctx *ctx, int64_t m, int iters, double *ms_per_launch);

/* direct access to the dense kernels for unit tests (device pointers; A is n x n col-major, ld) */
int spp_dense_potrf_upper(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld);
/* A x = b by the dense path (upper triangle of A read; A <- R, b <- x) */
int spp_dense_posv(spp_ctx *ctx, double *d_A, int64_t n, int64_t ld, double *d_b);
/* C (m x n, ldc) -= A^T B with A: k x m (lda), B: k x n (ldb): the MFMA trailing-update kernel */
int spp_dense_gemm_tn_sub(spp_ctx *ctx, int64_t m, int64_t n, int64_t k,
	const double *d_A, int64_t lda, const double *d_B, int64_t ldb, double *d_C, int64_t ldc);

const char *spp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPP_HIP_H */
