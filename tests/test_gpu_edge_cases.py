"""Edge cases of the boundary, through the C ABI on the GPU: degenerate sizes, ragged block widths, non-finite
input, malformed structures, structure changes between solves -- the situations the reference's solvers meet
through CNonlinearSolver_Lambda (a one-vertex system on the first incremental step, a failed factorization that
must leave the estimate alone, Clear_SymbolicDecomposition() when the graph grows)."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from slam_plus_plus_amd.blockcsc import structure_from_pairs
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu


def _spd_blocks(dim, pairs, seed):
    """random SPD block matrix with the given upper block pattern (dense construction, small sizes)"""
    rng = np.random.default_rng(seed)
    dim = np.asarray(dim, dtype=np.int32)
    base = np.concatenate([[0], np.cumsum(dim)]).astype(np.int64)
    n = int(base[-1])
    L = np.eye(n)
    for (i, j) in pairs:                      # one random "edge" per block pair: L += [Ji Jj]^T [Ji Jj]
        J = np.zeros((4, n))
        J[:, base[i]:base[i + 1]] = rng.normal(size=(4, dim[i]))
        J[:, base[j]:base[j + 1]] = rng.normal(size=(4, dim[j]))
        L += J.T @ J
    rows = np.array([min(p) for p in pairs], dtype=np.int64)
    cols = np.array([max(p) for p in pairs], dtype=np.int64)
    st, _, _ = structure_from_pairs(dim, rows, cols)
    vals = np.zeros(st.nvals)
    for j in range(st.nb):
        for p in range(st.col_ptr[j], st.col_ptr[j + 1]):
            i = st.row_idx[p]
            vals[st.blk_off[p]:st.blk_off[p] + int(dim[i]) * int(dim[j])] = L[base[i]:base[i + 1], base[j]:base[j + 1]].ravel(order="F")
    return st.with_vals(vals), L, rng.normal(size=n)


@pytest.mark.parametrize("dim,pairs", [([3], []), ([6], []), ([3, 3], [(0, 1)]), ([2, 5, 1, 4, 6, 3], [(0, 1), (1, 2), (0, 5), (3, 4), (2, 4)]),
                                       ([3] * 7, []), ([6, 3, 3], [(0, 1), (0, 2)]),
                                       ([7] * 12, [(i, i + 1) for i in range(11)] + [(0, 6), (2, 9), (4, 11)]),
                                       ([7, 3, 8, 7, 3, 3, 7], [(0, 1), (0, 3), (1, 3), (2, 3), (3, 4), (3, 6), (5, 6)])])
def test_tiny_ragged_and_disconnected_systems(dim, pairs):
    """one vertex, two vertices, block widths 1..8 mixed (7 = Sim(3) poses, Sim3_Types.h), only diagonal blocks (a
    disconnected graph), the smallest BA"""
    lam, L, eta = _spd_blocks(dim, pairs, 3)
    want = np.linalg.solve(L, eta)
    for mode in (api.MODE_AUTO, api.MODE_SPARSE):
        x = eta.copy()
        assert api.CLinearSolver_HIP(mode=mode).Solve_PosDef_Blocky(lam, x)
        assert np.linalg.norm(x - want) <= 1e-12 * np.linalg.norm(want)


@pytest.mark.parametrize("name,mode", [("se2_small", api.MODE_SPARSE), ("ba_small", api.MODE_SCHUR), ("ba_small", api.MODE_SCHUR_SPARSE)])
def test_non_finite_input_fails_cleanly(name, mode):
    """a NaN in Lambda: the factorization is reported as failed (Solve returns false), nothing hangs, the right-hand side
    is left alone -- what CNonlinearSolver_Lambda relies on to keep the estimate (NonlinearSolver_Lambda.h:628-664)"""
    lam, eta = orc.assemble(synth.make(name))
    vals = lam.vals.copy()
    p = lam.col_ptr[lam.nb // 2 + 1] - 1            # a diagonal block in the middle
    vals[lam.blk_off[p]] = np.nan
    solver = api.CLinearSolver_HIP(mode=mode)
    x = eta.copy()
    assert solver.Solve_PosDef_Blocky(lam.with_vals(vals), x) is False
    assert np.array_equal(x, eta)
    # and the same solver object recovers on the next, healthy system
    assert solver.Solve_PosDef_Blocky(lam, x)
    assert np.linalg.norm(lam.matvec(x) - eta) <= 1e-9 * np.linalg.norm(eta)


def test_structure_change_between_solves_reanalyzes():
    """Clear_SymbolicDecomposition() semantics: a solver object that has solved one structure is handed a bigger graph"""
    solver = api.CLinearSolver_HIP()
    for name in ("se2_small", "ba_small", "se3_small", "ba_tiny", "se2_small"):
        lam, eta = orc.assemble(synth.make(name))
        x = eta.copy()
        solver.Clear_SymbolicDecomposition()
        assert solver.Solve_PosDef_Blocky(lam, x)
        assert np.linalg.norm(lam.matvec(x) - eta) <= 1e-9 * np.linalg.norm(eta)


def test_malformed_structures_are_rejected_not_executed(hip_ctx):
    lib, h = hip_ctx.lib, hip_ctx.h
    lam, _, _ = _spd_blocks([3, 3, 3], [(0, 1), (1, 2)], 1)
    P = api._ptr

    def analyze(col_ptr, row_idx, blk_off, dim):
        return lib.spp_analyze(h, len(dim), P(np.asarray(col_ptr, dtype=np.int64)), P(np.asarray(row_idx, dtype=np.int64)),
                               P(np.asarray(blk_off, dtype=np.int64)), P(np.asarray(dim, dtype=np.int32)), api.MODE_AUTO)
    assert analyze(lam.col_ptr, lam.row_idx, lam.blk_off, lam.dim) == 0
    bad = lam.row_idx.copy()
    bad[lam.col_ptr[1]] = 2                                            # a block below the diagonal
    assert analyze(lam.col_ptr, bad, lam.blk_off, lam.dim) < 0
    assert analyze(lam.col_ptr, lam.row_idx, lam.blk_off, [3, 9, 3]) < 0   # block width outside 1..8
    nodiag_cp, nodiag_ri = [0, 1, 2, 3], [0, 0, 1]                         # columns 1, 2 without their diagonal block
    assert analyze(nodiag_cp, nodiag_ri, [0, 9, 18], [3, 3, 3]) < 0
    neg = lam.blk_off.copy()
    neg[0] = -9
    assert analyze(lam.col_ptr, lam.row_idx, neg, lam.dim) < 0
    assert "spp" in hip_ctx.last_error().lower() or hip_ctx.last_error()   # a message is available
    # the context is still usable
    assert analyze(lam.col_ptr, lam.row_idx, lam.blk_off, lam.dim) == 0
