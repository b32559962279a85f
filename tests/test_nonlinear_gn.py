"""The Gauss-Newton loop glue (slam_plus_plus_amd/nonlinear.py, mirror of
CNonlinearSolver_Lambda::Optimize, include/slam/NonlinearSolver_Lambda.h:539-666) against a golden
vector produced by the REFERENCE's own loop (tests/golden/se2_gn_400.npz, tools/make_golden_gn.py:
Optimize(5, 0.01) with CLinearSolver_UberBlock on a 400-pose / 589-edge graph).

CPU: the loop runs with the oracle as linear-solver path (test-only injection) -- checks linearization,
update, angle clamp, stopping rule. GPU: the product path (device assembly + HIP solve)."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import nonlinear
from oracle import spp_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "se2_gn_400.npz")
GOLD3 = os.path.join(os.path.dirname(__file__), "golden", "se3_gn_240.npz")   # 6 rings x 40 poses, 345 edges (dump3)


def _system():
    g = np.load(GOLD)
    info = np.tile(np.diag(g["info_diag"]), (g["edges"].shape[0], 1, 1))
    return nonlinear.CPoseGraph2D(g["init"], g["edges"], info), g


def _system3():
    g = np.load(GOLD3)
    info = np.tile(np.diag(g["info_diag"]), (g["edges"].shape[0], 1, 1))
    return nonlinear.CPoseGraph3D(g["init"], g["edges"], info), g


def _check3(system, g, solver):
    """3D: the reference linearizes with forward differences (delta = 1e-9, ~1e-7 relative noise in J), this code
    analytically. The graph is anchored only by the unit unary factor on vertex 0, so that noise moves the
    iterates along the nearly free rigid motion of the whole graph (0.14 m at the far end here) -- the
    comparison is therefore made on gauge-invariant quantities: every edge's residual at the final estimate
    and the objective."""
    from slam_plus_plus_amd.formats import se3_linearize
    assert solver.n_iterations <= int(g["max_iter"])
    r_mine = se3_linearize(system.poses, system.edges, system.info).r
    ref = nonlinear.CPoseGraph3D(g["final"], system.edges, system.info)
    r_ref = se3_linearize(ref.poses, ref.edges, ref.info).r
    assert np.abs(r_mine - r_ref).max() <= 2e-4, np.abs(r_mine - r_ref).max()
    assert abs(system.chi2() - ref.chi2()) <= 1e-4 * ref.chi2()


class _OraclePath:
    def solve(self, prob, first):
        lam, eta = orc.assemble(prob)
        st, x = orc.solve_blocky(lam, eta)
        return st == 0, x


def _check(system, g, solver):
    # the reference ran 5 iterations at most with threshold 0.01: same count, same states
    assert solver.n_iterations <= int(g["max_iter"])
    d = np.abs(system.poses - g["final"]).max()
    assert d <= 1e-6 * max(1.0, np.abs(g["final"]).max()), d


def test_gn_loop_glue_matches_the_reference_loop_cpu():
    system, g = _system()
    chi0 = system.chi2()
    solver = nonlinear.CNonlinearSolver_Lambda(system, path=_OraclePath())
    solver.Optimize(int(g["max_iter"]), float(g["threshold"]))
    _check(system, g, solver)
    assert system.chi2() < 0.05 * chi0


def test_gn_loop_glue_matches_the_reference_loop_3d_cpu():
    system, g = _system3()
    chi0 = system.chi2()
    solver = nonlinear.CNonlinearSolver_Lambda(system, path=_OraclePath())
    solver.Optimize(int(g["max_iter"]), float(g["threshold"]))
    _check3(system, g, solver)
    assert system.chi2() < 0.05 * chi0


def test_gn_stops_without_applying_a_step_below_the_threshold():
    system, g = _system()
    system.poses = g["final"].copy()
    before = system.poses.copy()
    solver = nonlinear.CNonlinearSolver_Lambda(system, path=_OraclePath())
    # at the reference's optimum the next step is below a loose threshold: one solve, no update
    assert solver.Optimize(5, 0.5) == 1
    assert np.array_equal(system.poses, before)
    assert 0 < solver.last_dx_norm <= 0.5


def test_gn_failed_factorization_leaves_the_estimate_unchanged():
    system, g = _system()

    class _Fail:
        def solve(self, prob, first):
            return False, None
    before = system.poses.copy()
    solver = nonlinear.CNonlinearSolver_Lambda(system, path=_Fail())
    assert solver.Optimize(5, 0.01) == 1
    assert np.array_equal(system.poses, before)


@pytest.mark.gpu
@pytest.mark.parametrize("host_jacobians", [True, False])
def test_gn_on_the_device_matches_the_reference_loop(host_jacobians):
    """host_jacobians=False: the whole iteration in HBM (device linearization, assembly, solve, update)"""
    system, g = _system()
    solver = nonlinear.CNonlinearSolver_Lambda(system, host_jacobians=host_jacobians)
    solver.Optimize(int(g["max_iter"]), float(g["threshold"]))
    _check(system, g, solver)
    solver.path.close()


@pytest.mark.gpu
@pytest.mark.parametrize("host_jacobians", [True, False])
def test_gn_on_the_device_matches_the_reference_loop_3d(host_jacobians):
    system, g = _system3()
    solver = nonlinear.CNonlinearSolver_Lambda(system, host_jacobians=host_jacobians)
    solver.Optimize(int(g["max_iter"]), float(g["threshold"]))
    _check3(system, g, solver)
    solver.path.close()


@pytest.mark.gpu
def test_device_linearization_matches_the_host_one():
    """spp_se2_linearize_device against formats.se2_linearize (analytic Jacobians of 2DSolverBase.h:373-418,
    as the reference's are): rounding-level agreement, including angle errors across the +-pi cut"""
    from slam_plus_plus_amd import api
    from slam_plus_plus_amd.formats import se2_linearize
    system, g = _system()
    rng = np.random.default_rng(5)
    system.poses[:, 2] += rng.uniform(-7, 7, size=system.poses.shape[0])   # headings beyond +-2 pi as well
    prob = se2_linearize(system.poses, system.edges, system.info)
    ctx = api.Context(0)
    ne = system.edges.shape[0]
    d = dict(v0=api.DeviceArray.from_host(ctx, prob.v0.astype(np.int32)), v1=api.DeviceArray.from_host(ctx, prob.v1.astype(np.int32)),
             poses=api.DeviceArray.from_host(ctx, system.poses.ravel()),
             meas=api.DeviceArray.from_host(ctx, np.ascontiguousarray(system.edges[:, 2:5]).ravel()),
             J0=api.DeviceArray(ctx, 9 * ne), J1=api.DeviceArray(ctx, 9 * ne), r=api.DeviceArray(ctx, 3 * ne))
    ctx.se2_linearize_device(ne, d["v0"].ptr, d["v1"].ptr, d["poses"].ptr, d["meas"].ptr, d["J0"].ptr, d["J1"].ptr, d["r"].ptr)
    ctx.synchronize()
    assert np.allclose(d["J0"].download().reshape(ne, 9), prob.J0, rtol=1e-13, atol=1e-13)
    assert np.allclose(d["J1"].download().reshape(ne, 9), prob.J1, rtol=1e-13, atol=1e-13)
    rd = d["r"].download().reshape(ne, 3)
    assert np.allclose(rd[:, :2], prob.r[:, :2], rtol=1e-12, atol=1e-12)
    # the angle error agrees modulo the representation of +-pi
    da = np.abs(rd[:, 2] - prob.r[:, 2])
    assert np.all(np.minimum(da, np.abs(da - 2 * np.pi)) < 1e-12) and np.all(np.abs(rd[:, 2]) <= np.pi + 1e-12)
    # update kernel: norm and (+) with the angle clamp
    dx = rng.normal(size=system.poses.size)
    ddx = api.DeviceArray.from_host(ctx, dx)
    assert abs(ctx.se2_update_device(system.poses.shape[0], d["poses"].ptr, ddx.ptr, apply=False) - np.linalg.norm(dx)) < 1e-12 * np.linalg.norm(dx)
    assert np.array_equal(d["poses"].download(), system.poses.ravel())
    ctx.se2_update_device(system.poses.shape[0], d["poses"].ptr, ddx.ptr, apply=True)
    want = system.poses + dx.reshape(-1, 3)
    want[:, 2] = np.fmod(want[:, 2], 2 * np.pi)
    assert np.allclose(d["poses"].download().reshape(-1, 3), want, rtol=0, atol=1e-14)
    ctx.close()
