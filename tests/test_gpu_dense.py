"""GPU parity of the dense kernels (MFMA f64 trailing update, blocked potrf, solves) against
numpy fp64. Tolerances are for fp64 arithmetic: 1e-12 relative unless noted."""
import numpy as np
import pytest

from slam_plus_plus_amd import api

pytestmark = pytest.mark.gpu


def _gemm(ctx, m, n, k, seed):
    rng = np.random.default_rng(seed)
    lda, ldb, ldc = k + 2, k + 4, m + 3
    A = rng.standard_normal((k, m))
    B = rng.standard_normal((k, n))
    C = rng.standard_normal((m, n))
    Af = np.zeros((lda, m), order="F"); Af[:k] = A
    Bf = np.zeros((ldb, n), order="F"); Bf[:k] = B
    Cf = np.zeros((ldc, n), order="F"); Cf[:m] = C
    dA = api.DeviceArray.from_host(ctx, Af.ravel(order="F"))
    dB = api.DeviceArray.from_host(ctx, Bf.ravel(order="F"))
    dC = api.DeviceArray.from_host(ctx, Cf.ravel(order="F"))
    ctx._check(ctx.lib.spp_dense_gemm_tn_sub(ctx.h, m, n, k, dA.ptr, lda, dB.ptr, ldb, dC.ptr, ldc))
    out = dC.download().reshape((ldc, n), order="F")[:m]
    for d in (dA, dB, dC):
        d.free()
    return out, C - A.T @ B


@pytest.mark.parametrize("m,n,k", [(16, 16, 16), (64, 64, 32), (128, 128, 128), (100, 37, 48), (300, 520, 128),
                                   (2048, 2048, 128), (129, 1, 16)])
def test_gemm_tn_sub_matches_numpy(hip_ctx, m, n, k):
    """asymmetric random operands: a swapped MFMA row/col map cannot pass"""
    got, want = _gemm(hip_ctx, m, n, k, 7 + m + n)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err < 1e-13, err


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    return M @ M.T / n + np.eye(n) * 2.0


@pytest.mark.parametrize("n", [1, 5, 127, 128, 129, 294, 640, 1000])
def test_potrf_upper_reconstructs(hip_ctx, n):
    A = _spd(n, n)
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(A).ravel(order="F"))
    st = hip_ctx._check(hip_ctx.lib.spp_dense_potrf_upper(hip_ctx.h, dA.ptr, n, n))
    assert st == 0
    R = np.triu(dA.download().reshape((n, n), order="F"))
    dA.free()
    err = np.abs(R.T @ R - A).max() / np.abs(A).max()
    assert err < 1e-13, err
    Rref = np.linalg.cholesky(A).T
    assert np.abs(R - Rref).max() / np.abs(Rref).max() < 1e-11


@pytest.mark.parametrize("n", [3, 128, 294, 777, 2000])
def test_posv_matches_numpy(hip_ctx, n):
    A = _spd(n, 100 + n)
    b = np.random.default_rng(n).standard_normal(n)
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(A).ravel(order="F"))
    db = api.DeviceArray.from_host(hip_ctx, b)
    st = hip_ctx._check(hip_ctx.lib.spp_dense_posv(hip_ctx.h, dA.ptr, n, n, db.ptr))
    assert st == 0
    x = db.download()
    dA.free(); db.free()
    xr = np.linalg.solve(A, b)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-11


@pytest.mark.parametrize("n", [2688, 3072, 3201, 4224, 5800])
def test_posv_large_sizes(hip_ctx, n):
    """sizes that run the two-stream schedule with the mixed-granularity trailing update (whole 128 x 128 tiles +
    64 x 64 quarters), its rectangular right-hand-side tile column (n a multiple of 128) and the hand-over to
    the single-stream steps; up to 44 tile rows the whole factorization is the streamed launch (spp_dense_tail.h), at
    5800 (46 tile rows) the two-stream schedule does the first steps and hands the rest to it"""
    A = _spd(n, 7 + n)
    b = np.random.default_rng(n).standard_normal(n)
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(A).ravel(order="F"))
    db = api.DeviceArray.from_host(hip_ctx, b)
    st = hip_ctx._check(hip_ctx.lib.spp_dense_posv(hip_ctx.h, dA.ptr, n, n, db.ptr))
    assert st == 0
    x = db.download()
    R = np.triu(dA.download().reshape((n, n), order="F"))
    dA.free(); db.free()
    xr = np.linalg.solve(A, b)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-11
    assert np.abs(R.T @ R - A).max() / np.abs(A).max() < 1e-13


def test_streamed_factor_reports_a_late_non_positive_pivot(hip_ctx):
    """the streamed factorization (spp_dense_tail.h; 24 tile rows = 300 tiles, more than the CUs hold at once): a
    diagonal tile deep inside fails, raises the abort word, every workgroup waiting for its row tiles gives up --
    the call returns NOT_POSDEF (reference contract: BlockMatrix.cpp:9765-9771) instead of timing out, and the next
    factorization on the same context is unaffected"""
    n = 3000
    A = _spd(n, 11)
    B = A.copy()
    B[2500, 2500] = -1.0
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(B).ravel(order="F"))
    st = hip_ctx.lib.spp_dense_potrf_upper(hip_ctx.h, dA.ptr, n, n)
    assert st == api.SPP_NOT_POSDEF
    dA.free()
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(A).ravel(order="F"))
    st = hip_ctx._check(hip_ctx.lib.spp_dense_potrf_upper(hip_ctx.h, dA.ptr, n, n))
    assert st == 0
    R = np.triu(dA.download().reshape((n, n), order="F"))
    dA.free()
    assert np.abs(R.T @ R - A).max() / np.abs(A).max() < 1e-13


def test_potrf_reports_not_posdef(hip_ctx):
    """reference contract: non-positive pivot -> false (BlockMatrix.cpp:9765-9771)"""
    n = 200
    A = _spd(n, 5)
    A[150, 150] = -1.0
    dA = api.DeviceArray.from_host(hip_ctx, np.asfortranarray(A).ravel(order="F"))
    st = hip_ctx.lib.spp_dense_potrf_upper(hip_ctx.h, dA.ptr, n, n)
    dA.free()
    assert st == api.SPP_NOT_POSDEF


def test_backward_chain_with_more_block_rows_than_compute_units(hip_ctx):
    """n = 33 000: 258 block rows of 128, more than the 256 CUs -- and the single-launch backward substitution holds a
    whole CU per workgroup, so not all of its workgroups can be resident. Progress must not depend on residency
    (block row = last minus blockIdx: producers are dispatched before their consumers). Diagonal + rank-8 matrix
    (8.7 GB); checked through the residual."""
    n = 33000
    rng = np.random.default_rng(5)
    V = rng.standard_normal((n, 8))
    A = V @ V.T
    A[np.diag_indices(n)] += 4.0 + rng.random(n)
    b = rng.standard_normal(n)
    dA = api.DeviceArray.from_host(hip_ctx, A.reshape(-1))   # symmetric: row-major == column-major
    db = api.DeviceArray.from_host(hip_ctx, b)
    st = hip_ctx._check(hip_ctx.lib.spp_dense_posv(hip_ctx.h, dA.ptr, n, n, db.ptr))
    assert st == 0
    x = db.download()
    dA.free(); db.free()
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    assert res < 1e-12, res
