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


def test_config5_full_size_properties():
    """BASELINE.json config 5 at full size (10 000 cameras, 2 000 000 points, 10 000 000 observations),
    entirely on the device: assembly, analysis (AUTO must keep the 60 000-unknown reduced system
    sparse and pick nested dissection: a handful of tree levels, not thousands), solve.
    Size-independent properties: tiny residual of the full system, bit-reproducibility, and the
    damped solution shrinking when the damping grows."""
    prob = synth.make("synthetic10k")
    ctx = api.Context(0, api.FLAG_PROFILE)
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    assert (st.nb, prob.v0.size) == (10000 + 2000000, 10000000)
    arrs = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    dv, de, dr = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n), api.DeviceArray(ctx, st.n)
    ctx.assemble_device(*[a.ptr for a in arrs], prob.damping, dv.ptr, de.ptr)
    ctx.analyze(st, api.MODE_AUTO)
    assert ctx.info("MODE") == api.MODE_SCHUR_SPARSE and ctx.info("N_REDUCED") == 60000
    assert ctx.info("N_LEVELS") < 64, "the elimination tree of the trajectory must not be a chain"
    assert ctx.info("FACTOR_NNZ") < 0.01 * 60000 ** 2 / 2
    xs = []
    for _ in range(2):
        dr.copy_from(de)
        assert ctx.factor_solve_device(dv.ptr, dr.ptr) == 0
        xs.append(dr.download())
    assert np.array_equal(xs[0], xs[1]), "bit-reproducible"
    lam = st.with_vals(dv.download())
    eta = de.download()
    res = np.linalg.norm(lam.matvec(xs[0]) - eta) / np.linalg.norm(eta)
    assert res < 1e-12, res
    ctx.assemble_device(*[a.ptr for a in arrs], 10 * prob.damping, dv.ptr, de.ptr)
    dr.copy_from(de)
    assert ctx.factor_solve_device(dv.ptr, dr.ptr) == 0
    assert np.linalg.norm(dr.download()) < np.linalg.norm(xs[0])
    for d in arrs + [dv, de, dr]:
        d.free()
    ctx.close()
