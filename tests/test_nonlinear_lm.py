"""The Levenberg-Marquardt loop (slam_plus_plus_amd/nonlinear.py: CNonlinearSolver_Lambda_LM, mirror of
include/slam/NonlinearSolver_Lambda_LM.h:796-1135 with the damping policy of :151-222) against a golden
vector produced by the REFERENCE's own LM solver (tests/golden/ba_lm_12.npz, tools/make_golden_gn.py lm:
CNonlinearSolver_Lambda_LM + CLinearSolver_Schur/UberBlock, Optimize(5, 0.01), 12 cameras / 300 points /
1368 observations).

The reference differentiates the projection numerically (delta = 1e-9), this code analytically: the final
objective agrees to 1e-6 relative, the states to the few 1e-5 that noise leaves along the weakly determined
directions of a BA problem.
CPU: loop glue with a numpy + oracle path (test-only injection). GPU: the resident device path."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import nonlinear
from oracle import spp_oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ba_lm_12.npz"))


class _HostBAPath:
    def begin(self, s):
        self.s = s

    def linearize(self):
        self.prob = self.s.linearize()

    def max_hessian_diag(self):
        p = self.prob
        return max((p.J0.reshape(-1, 6, 2) ** 2).sum(2).max(), (p.J1.reshape(-1, 3, 2) ** 2).sum(2).max())

    def chi2(self):
        self.linearize()
        return float((self.prob.r ** 2).sum())

    def solve(self, alpha):
        self.prob["damping"] = alpha
        lam, eta = orc.assemble(self.prob)
        st, x, _ = orc.schur_solve(lam, eta)
        self.dx, self.eta = x, eta
        return st == 0, float(np.linalg.norm(x)) if st == 0 else 0.0

    def gain_denominator(self, alpha):
        return float(self.dx @ (alpha * self.dx + self.eta))

    def save(self):
        self.saved = self.s.state()

    def restore(self):
        self.s.set_state(self.saved)

    def apply(self):
        self.s.plus(self.dx)

    def finish(self, s):
        pass


def _system():
    return nonlinear.CBundleAdjustment(G["cams"], G["intr"], G["points"], G["obs"])


def _check(system, solver):
    from scipy.spatial.transform import Rotation
    ref = nonlinear.CBundleAdjustment(G["final_cams"], G["intr"], G["final_points"], G["obs"])
    assert abs(system.chi2() - ref.chi2()) <= 1e-6 * ref.chi2(), (system.chi2(), ref.chi2())
    assert np.abs(system.cams[:, :3] - ref.cams[:, :3]).max() < 2e-4
    assert (Rotation.from_rotvec(system.cams[:, 3:]) * Rotation.from_rotvec(ref.cams[:, 3:]).inv()).magnitude().max() < 5e-6
    assert np.abs(system.points - ref.points).max() < 1e-4
    assert solver.chi2_history[-1] < 0.1 * solver.chi2_history[0]


def test_lm_loop_glue_matches_the_reference_lm_cpu():
    system = _system()
    solver = nonlinear.CNonlinearSolver_Lambda_LM(system, path=_HostBAPath())
    solver.Optimize(int(G["max_iter"]), float(G["threshold"]))
    _check(system, solver)


def test_lm_rejects_a_rising_step_and_restores_the_state():
    """a path whose objective rises after the update: the step is rolled back, the damping grows by nu = 2, 4, ..."""
    system = _system()

    class _Rising(_HostBAPath):
        def chi2(self):
            v = super().chi2()
            self.calls = getattr(self, "calls", 0) + 1
            return v if self.calls == 1 else v + 1e9     # every evaluation after the initial one looks worse
    before = system.state()
    solver = nonlinear.CNonlinearSolver_Lambda_LM(system, path=_Rising())
    n = solver.Optimize(2, 0.0)
    assert n == 2 + 10                                   # the iteration budget grows once per failure, ten times at most
    assert np.array_equal(system.cams, before[0]) and np.array_equal(system.points, before[1])
    assert solver.alpha > 1e15                           # alpha0 (~30) times 2 * 4 * 8 * ... over twelve rejections


@pytest.mark.gpu
def test_lm_on_the_device_matches_the_reference_lm():
    system = _system()
    solver = nonlinear.CNonlinearSolver_Lambda_LM(system)      # linearization, assembly, solve, control scalars: all in HBM
    solver.Optimize(int(G["max_iter"]), float(G["threshold"]))
    _check(system, solver)
    solver.path.close()
