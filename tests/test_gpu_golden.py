"""GPU: the HIP path against the committed golden vectors the REFERENCE produced (tests/golden),
without the reference build being needed on the box, plus size-independent properties at the full
BASELINE.json sizes (Venice-871-shaped BA)."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# pose graphs anchored by one unit prior (cond 1e7 .. 1e11): direct bound on ||dx - dx_ref|| / ||dx_ref|| per fixture
DIRECT_BOUND = {"se2_small": 2e-8, "se3_small": 2e-8, "manhattan3500": 1e-6, "sphere2500": 5e-9}


@pytest.mark.parametrize("name", ["ba_tiny", "ba_small", "ba_interleaved", "ladybug49", "se2_small", "se3_small",
                                  "manhattan3500", "sphere2500"])
def test_hip_solution_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    s = api.CLinearSolver_HIP()
    x = eta.copy()
    assert s.Solve_PosDef_Blocky(lam, x)
    stride = int(g["dx_stride"])
    if name.startswith(("ba", "lady")):
        # LM-damped BA systems (cond ~1e5): north_star's bound against every reference backend directly
        for key in g.files:
            if key.startswith("dx_") and key != "dx_stride":
                assert _rel(x[::stride], g[key]) < 1e-10, (key, _rel(x[::stride], g[key]))
    # every case: at most 4x less accurate than the reference is itself, both measured against the refined solution
    # of the same Lambda (tests/parity.py; on the pose graphs the reference's own backends are 1e-9..1e-7 away from it)
    import parity
    parity.check_against_reference(x, g, lam, eta)
    # ... and the plain relative difference to every stored reference solution, with an explicit bound per fixture (about
    # three times what was measured on MI355X: manhattan3500 3.0e-7, sphere2500 1.4e-9 against CLinearSolver_UberBlock): a
    # regression that stays inside the yardstick's 4x slack still shows here
    if name in DIRECT_BOUND:
        for key in g.files:
            if key.startswith("dx_") and key != "dx_stride":
                assert _rel(x[::stride], g[key]) < DIRECT_BOUND[name], (key, _rel(x[::stride], g[key]))


def test_venice_full_size_properties():
    """BASELINE.json config 4 at full size, entirely on the device: assembly, analysis, solve.
    Properties that do not need a CPU solve: tiny residual of the full system, determinism, and the
    solution of the damped system shrinking when the damping grows."""
    prob = synth.make("venice871")
    ctx = api.Context(0, api.FLAG_PROFILE)
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    assert (st.nb, prob.v0.size) == (871 + 530304, 2838740)
    arrs = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    dv, de, dr = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n), api.DeviceArray(ctx, st.n)
    ctx.assemble_device(*[a.ptr for a in arrs], prob.damping, dv.ptr, de.ptr)
    ctx.analyze(st, api.MODE_AUTO)
    assert ctx.info("MODE") == api.MODE_SCHUR and ctx.info("N_REDUCED") == 5226
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
