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


def _system():
    g = np.load(GOLD)
    info = np.tile(np.diag(g["info_diag"]), (g["edges"].shape[0], 1, 1))
    return nonlinear.CPoseGraph2D(g["init"], g["edges"], info), g


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
def test_gn_on_the_device_matches_the_reference_loop():
    system, g = _system()
    solver = nonlinear.CNonlinearSolver_Lambda(system)   # device assembly + HIP solve
    solver.Optimize(int(g["max_iter"]), float(g["threshold"]))
    _check(system, g, solver)
    solver.path.close()
