"""GPU parity of the Lambda / eta assembly kernels against the CPU oracle's restatement of
Calculate_Hessians_v2 + ReduceAll. Low-degree vertices are summed in the reference's edge order
(same order; the device contracts a*b+c into FMAs, so values agree to an ulp, not bitwise); high-degree vertices (cameras) use a fixed butterfly order, so the bound is a few
ulps relative: 1e-13."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu


def _assemble_gpu(ctx, prob, damping):
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    dJ0 = api.DeviceArray.from_host(ctx, prob.J0.ravel())
    dJ1 = api.DeviceArray.from_host(ctx, prob.J1.ravel())
    dOm = api.DeviceArray.from_host(ctx, prob.Om.ravel())
    dr = api.DeviceArray.from_host(ctx, prob.r.ravel())
    dv = api.DeviceArray(ctx, st.nvals)
    de = api.DeviceArray(ctx, st.n)
    ctx.assemble_device(dJ0.ptr, dJ1.ptr, dOm.ptr, dr.ptr, damping, dv.ptr, de.ptr)
    vals, eta = dv.download(), de.download()
    for d in (dJ0, dJ1, dOm, dr, dv, de):
        d.free()
    return st.with_vals(vals), eta


@pytest.mark.parametrize("name", ["ba_tiny", "ba_small", "ba_interleaved", "ba_medium", "se2_small", "se3_small",
                                  "manhattan3500", "sphere2500", "ladybug49", "lm2d_small", "lm2d_interleaved"])
def test_assembly_matches_oracle(hip_ctx, name):
    prob = synth.make(name)
    lam_o, eta_o = orc.assemble(prob)
    lam_g, eta_g = _assemble_gpu(hip_ctx, prob, prob.damping)
    assert np.array_equal(lam_g.col_ptr, lam_o.col_ptr)
    assert np.array_equal(lam_g.row_idx, lam_o.row_idx)
    assert np.array_equal(lam_g.blk_off, lam_o.blk_off)
    scale = np.abs(lam_o.vals).max()
    assert np.abs(lam_g.vals - lam_o.vals).max() <= 1e-13 * scale
    assert np.abs(eta_g - eta_o).max() <= 1e-13 * max(1.0, np.abs(eta_o).max())


def test_assembled_lambda_feeds_the_solver(hip_ctx):
    """assemble -> analyze -> solve entirely on the device (the GN iteration of bench.py)"""
    prob = synth.make("ba_medium")
    lam_o, eta_o = orc.assemble(prob)
    st = hip_ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    arrs = [api.DeviceArray.from_host(hip_ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    dv = api.DeviceArray(hip_ctx, st.nvals)
    de = api.DeviceArray(hip_ctx, st.n)
    hip_ctx.assemble_device(*[a.ptr for a in arrs], prob.damping, dv.ptr, de.ptr)
    hip_ctx.analyze(st, api.MODE_AUTO)
    assert hip_ctx.factor_solve_device(dv.ptr, de.ptr) == 0
    x = de.download()
    st_o, xo, _ = orc.schur_solve(lam_o, eta_o)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-10
    for d in arrs + [dv, de]:
        d.free()


CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
sys.path.insert(0, %r)
from test_gpu_assemble import _assemble_gpu
ctx = api.Context(0, 0)
for name in sys.argv[1:]:
    prob = synth.make(name)
    lam_o, eta_o = orc.assemble(prob)
    lam_g, eta_g = _assemble_gpu(ctx, prob, prob.damping)
    assert np.array_equal(lam_g.col_ptr, lam_o.col_ptr) and np.array_equal(lam_g.row_idx, lam_o.row_idx)
    assert np.array_equal(lam_g.blk_off, lam_o.blk_off)
    assert np.abs(lam_g.vals - lam_o.vals).max() <= 1e-13 * np.abs(lam_o.vals).max()
    assert np.abs(eta_g - eta_o).max() <= 1e-13 * max(1.0, np.abs(eta_o).max())
    ctx.analyze(lam_g, api.MODE_AUTO)
    code, x = ctx.factor_solve(lam_g.vals, eta_g)
    assert code == 0
    assert np.linalg.norm(lam_o.matvec(x) - eta_o) <= 1e-11 * np.linalg.norm(eta_o)
print("ok")
"""


def test_assembly_plan_cut_among_host_threads():
    """the threaded passes of spp_assemble_analyze / spp_analyze (edge sort with atomic increments, Lambda structure by
    ranges of columns, structure copy) normally start at 2^18 edges: SPP_PLAN_MIN_WORK=1 forces them on small graphs,
    in a process of its own (the switch is read once per process)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPP_PLAN_THREADS="7", SPP_PLAN_MIN_WORK="1")
    r = subprocess.run([sys.executable, "-c", CHILD % (root, os.path.join(root, "tests")), "ba_small", "ba_interleaved",
                        "se3_small", "lm2d_interleaved"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
