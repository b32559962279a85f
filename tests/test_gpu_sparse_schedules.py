"""The schedules of the sparse path give the same solution.

Default: the whole assembly tree in one dependency-driven launch per direction (front_dag_kernel / front_bwd_dag_kernel),
big fronts by teams of workgroups. Fallbacks, selected by environment variables that the library reads once per process
(hence one child process per variant, started before this process's own GPU work matters to it):
    SPP_SPARSE_TEAMS=0   big fronts and everything above them level by level through the host-driven dense factor
    SPP_SPARSE_DAG=0     one launch per level and size class (the round-1 schedule, also taken after a timed-out flag wait)
    SPP_DAG_SPLIT=2      the bottom of the tree (fronts of at most 64 rows) as a launch of its own with small workgroups --
                         taken by itself only from 1024 such fronts on
Every variant must reproduce the default's solution of the sphere2500-shaped system (big fronts up to 655 rows, 14 levels)
to the rounding of a different summation order, and each must be bit-reproducible run to run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
lam, eta = orc.assemble(synth.make("sphere2500"))
s = api.CLinearSolver_HIP(mode=api.MODE_SPARSE)
x = eta.copy(); assert s.Solve_PosDef_Blocky(lam, x)
y = eta.copy(); assert s.Solve_PosDef_Blocky(lam, y)
assert np.array_equal(x, y), "not bit-reproducible"
res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta)
assert res < 1e-11, res
np.save(sys.argv[1], x)
print("levels", s.ctx.info("N_LEVELS"), "residual", res)
"""


def _run(tmp_path, tag, env_extra):
    out = tmp_path / (tag + ".npy")
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out)


def test_fallback_schedules_match_the_dependency_driven_one(tmp_path):
    x_dag = _run(tmp_path, "dag", {})
    x_noteam = _run(tmp_path, "noteam", {"SPP_SPARSE_TEAMS": "0"})
    x_levels = _run(tmp_path, "levels", {"SPP_SPARSE_DAG": "0"})
    x_split = _run(tmp_path, "split", {"SPP_DAG_SPLIT": "2"})
    n = np.linalg.norm(x_dag)
    # cond ~1e11 on this undamped pose graph: solutions of different (equally valid) summation orders differ by ~cond * eps
    assert np.linalg.norm(x_noteam - x_dag) / n < 1e-6
    assert np.linalg.norm(x_levels - x_dag) / n < 1e-6
    assert np.array_equal(x_split, x_dag)  # the same fronts in the same order: only the launch boundaries differ


def test_timed_out_dag_launch_is_repeated_level_by_level(tmp_path):
    """SPP_DAG_TIMEOUT_TICKS=1 makes the first flag wait of the dependency-driven launch time out: the solve must not
    fail -- the right-hand side is left untouched by the aborted launch, the library repeats the solve level by level
    (and stays there) -- and must give exactly the level-by-level schedule's solution."""
    x_levels = _run(tmp_path, "levels2", {"SPP_SPARSE_DAG": "0"})
    x_abort = _run(tmp_path, "abort", {"SPP_DAG_TIMEOUT_TICKS": "1"})
    assert np.array_equal(x_abort, x_levels)
