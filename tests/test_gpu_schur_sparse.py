"""GPU parity of the Schur path with a SPARSE reduced camera system (SPP_MODE_SCHUR_SPARSE): landmark
elimination as in the dense variant, S kept as block-CSC and factored by the supernodal multifrontal
kernels -- the reference's CLinearSolver_Schur with a sparse inner solver
(include/slam/LinearSolver_Schur.h:1844-1853), the shape of BASELINE config 5 (10k cameras).

Checked against the dense-S variant of the same library, the CPU oracle and (when oracle/_ref is
present) the reference's own CLinearSolver_Schur. Tolerance: north_star's 1e-10 relative."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("name", ["ba_tiny", "ba_small", "ba_interleaved", "ba_banded", "ladybug49"])
def test_sparse_reduced_system_matches_dense_variant_oracle_and_reference(name):
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    sp = api.CLinearSolver_HIP(mode=api.MODE_SCHUR_SPARSE)
    x = eta.copy()
    assert sp.Solve_PosDef_Blocky(lam, x)
    assert sp.ctx.info("MODE") == api.MODE_SCHUR_SPARSE
    assert sp.ctx.info("S_LD") == 0 and sp.ctx.info("S_NNZB") > 0
    assert sp.ctx.info("N_REDUCED") == 6 * int((lam.dim == 6).sum())
    de = api.CLinearSolver_HIP(mode=api.MODE_SCHUR)
    xd = eta.copy()
    assert de.Solve_PosDef_Blocky(lam, xd)
    assert de.ctx.info("S_NNZB") == sp.ctx.info("S_NNZB")
    assert _rel(x, xd) < TOL, _rel(x, xd)
    st, xo, _ = orc.schur_solve(lam, eta)
    assert st == 0 and _rel(x, xo) < TOL
    res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta)
    assert res < 1e-12, res
    if orc.have_ref():
        rs = orc.RefSolver("schur", lam)
        st, xr, _ = rs.solve(lam.vals, eta)
        assert st == 0 and _rel(x, xr) < TOL, _rel(x, xr)
    # the ordering reported for the drop-in: a permutation of all block columns, poses (in the
    # elimination order of the reduced system) before landmarks
    order = sp.ctx.ordering(lam.nb)
    assert sorted(order.tolist()) == list(range(lam.nb))
    nc = int((lam.dim == 6).sum())
    assert np.all(lam.dim[order[:nc]] == 6) and np.all(lam.dim[order[nc:]] == 3)
    # same structure again: symbolic reuse, bit-reproducible
    x2 = eta.copy()
    assert sp.Solve_PosDef_Blocky(lam, x2)
    assert np.array_equal(x, x2)


def test_sparse_reduced_system_is_banded_and_small_on_the_trajectory_problem():
    prob = synth.make("ba_banded")
    lam, eta = orc.assemble(prob)
    sp = api.CLinearSolver_HIP(mode=api.MODE_SCHUR_SPARSE)
    x = eta.copy()
    assert sp.Solve_PosDef_Blocky(lam, x)
    nc = int((lam.dim == 6).sum())
    assert sp.ctx.info("S_NNZB") < 0.1 * nc * (nc + 1) / 2
    # the factor of the banded system stays far below the dense n_red^2 / 2
    assert sp.ctx.info("FACTOR_NNZ") < 0.25 * (6 * nc) ** 2 / 2


def test_sparse_variant_not_posdef_returns_false_and_keeps_eta():
    prob = synth.make("ba_small")
    lam, eta = orc.assemble(prob)
    vals = lam.vals.copy()
    cam = int(np.flatnonzero(lam.dim == 6)[3])
    p = lam.col_ptr[cam + 1] - 1
    vals[lam.blk_off[p]:lam.blk_off[p] + 36] *= -1.0
    bad = lam.with_vals(vals)
    solver = api.CLinearSolver_HIP(mode=api.MODE_SCHUR_SPARSE)
    x = eta.copy()
    assert solver.Solve_PosDef_Blocky(bad, x) is False
    assert np.array_equal(x, eta)
