"""GPU parity of the Schur (bundle adjustment) path through the C ABI against the CPU oracle and,
when oracle/_ref is present, the reference's own CLinearSolver_Schur / CHOLMOD / CSparse / UberBlock.

Tolerance: north_star's ||dx_gpu - dx_ref|| / ||dx_ref|| < 1e-10 (fp64). It is applied on
LM-damped BA systems (cond ~1e5), for which the reference's own backends agree to ~1e-13."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("name", ["ba_tiny", "ba_small", "ba_interleaved", "ba_medium", "ladybug49"])
def test_schur_solve_matches_oracle_and_reference(name):
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    solver = api.CLinearSolver_HIP(mode=api.MODE_AUTO)
    x = eta.copy()
    assert solver.Solve_PosDef_Blocky(lam, x)
    assert solver.ctx.info("MODE") == api.MODE_SCHUR
    st, xo, _ = orc.schur_solve(lam, eta)
    assert st == 0
    assert _rel(x, xo) < TOL
    # residual of the full system: size-independent property
    res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta)
    assert res < 1e-12, res
    if orc.have_ref():
        for be in ("schur", "uberblock", "cholmod"):
            rs = orc.RefSolver(be, lam)
            st, xr, _ = rs.solve(lam.vals, eta)
            assert st == 0
            assert _rel(x, xr) < TOL, (be, _rel(x, xr))
    # second call with the same structure reuses the symbolic analysis and is bit-reproducible
    x2 = eta.copy()
    assert solver.Solve_PosDef_Blocky(lam, x2)
    assert np.array_equal(x, x2)


def test_schur_not_posdef_returns_false_and_keeps_eta():
    prob = synth.make("ba_small")
    lam, eta = orc.assemble(prob)
    vals = lam.vals.copy()
    # make one camera block indefinite
    cam = int(np.flatnonzero(lam.dim == 6)[3])
    p = lam.col_ptr[cam + 1] - 1
    vals[lam.blk_off[p]:lam.blk_off[p] + 36] *= -1.0
    bad = lam.with_vals(vals)
    solver = api.CLinearSolver_HIP()
    x = eta.copy()
    assert solver.Solve_PosDef_Blocky(bad, x) is False
    assert np.array_equal(x, eta)


def test_device_resident_solve_and_phase_timers(hip_ctx):
    prob = synth.make("ba_medium")
    lam, eta = orc.assemble(prob)
    hip_ctx.analyze(lam, api.MODE_SCHUR)
    dv = api.DeviceArray.from_host(hip_ctx, lam.vals)
    dr = api.DeviceArray.from_host(hip_ctx, eta)
    assert hip_ctx.factor_solve_device(dv.ptr, dr.ptr) == 0
    x = dr.download()
    st, xo, _ = orc.schur_solve(lam, eta)
    assert _rel(x, xo) < TOL
    ph = hip_ctx.phase_ms()
    assert ph["total"] > 0 and ph["factor"] > 0
    dv.free(); dr.free()
