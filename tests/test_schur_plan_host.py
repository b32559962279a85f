"""Host-only: the symbolic Schur plan (spp_schur_plan_host -- guided ordering, observation lists, block pattern of S and
its per-block lists of block products; what CLinearSolver_Schur recomputes structurally in every call,
LinearSolver_Schur.cpp:771-838, LinearSolver_Schur.h:1699-1709, BlockMatrixFBS.inl:1147-1304). No GPU needed: the plan is
pure integer work on host threads; its result must not depend on their number."""
import os
import subprocess
import sys

import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json
sys.path.insert(0, %r)
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
lam = orc.lambda_structure(synth.make(sys.argv[1]))[0]
d = api.schur_plan_host(lam, int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1")
d.pop("seconds")
print(json.dumps(d))
"""


def _plan(name, threads, rank=0, world=1, sparse=False, min_work=None):
    import json
    env = dict(os.environ, SPP_PLAN_THREADS=str(threads))
    if min_work is not None:
        env["SPP_PLAN_MIN_WORK"] = str(min_work)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, name, str(rank), str(world), "1" if sparse else "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().split("\n")[-1])


@pytest.mark.parametrize("name", ["ba_small", "ba_interleaved", "ba_banded"])
def test_plan_counts_match_the_graph(name):
    prob = synth.make(name)
    lam = orc.lambda_structure(prob)[0]
    d = api.schur_plan_host(lam)
    dim = np.asarray(lam.dim)
    is_lm = dim == dim.min()
    assert d["nc"] == int((~is_lm).sum()) and d["nl"] == int(is_lm.sum())
    assert d["no"] == prob.v0.size
    # one block product per pair of observers of a landmark (a <= b): sum over landmarks of k (k + 1) / 2
    lm_of_edge = np.where(is_lm[prob.v0], prob.v0, prob.v1)
    k = np.bincount(lm_of_edge, minlength=dim.size)[is_lm]
    assert d["n_pairs"] == int((k * (k + 1) // 2).sum())
    # S holds every pose's diagonal block and at most every co-observed pair
    assert d["nc"] <= d["n_sblk"] <= d["nc"] * (d["nc"] + 1) // 2
    assert d["n_items"] >= d["n_sblk"] and d["n_multi"] <= d["n_sblk"]


def test_plan_does_not_depend_on_the_thread_count():
    # 100 000 observations: above the size from which the passes are cut among the threads
    a = _plan("ba_medium", 1)
    b = _plan("ba_medium", 4)
    c = _plan("ba_medium", 7)
    assert a == b == c, (a, b, c)


@pytest.mark.parametrize("name,rank,world,sparse", [("ba_small", 0, 1, False), ("ba_interleaved", 0, 1, False),
                                                   ("ba_medium", 1, 2, False), ("ba_banded", 1, 2, True)])
def test_every_pass_cut_among_threads_gives_the_same_plan(name, rank, world, sparse):
    """SPP_PLAN_MIN_WORK=1 cuts every pass (observation scan, camera lists, pair lists by row of S) among the threads
    even on a small graph: more threads than some ranges have columns, the unsorted-observation fallback of the
    interleaved numbering, a landmark shard, and the union pattern of a sharded sparse S"""
    a = _plan(name, 1, rank, world, sparse)
    b = _plan(name, 5, rank, world, sparse, min_work=1)
    c = _plan(name, 16, rank, world, sparse, min_work=1)
    assert a == b == c, (a, b, c)


def test_landmark_shards_partition_the_work():
    full = _plan("ba_medium", 2)
    shards = [_plan("ba_medium", 2, r, 2) for r in range(2)]
    assert sum(s["nl"] for s in shards) == full["nl"] and abs(shards[0]["nl"] - shards[1]["nl"]) <= 1
    assert sum(s["no"] for s in shards) == full["no"]
    assert sum(s["n_pairs"] for s in shards) == full["n_pairs"]
    assert all(s["nc"] == full["nc"] for s in shards)
    # sparse reduced system: every rank holds the UNION block structure (the all-reduced buffer adds like blocks)
    u = [_plan("ba_banded", 2, r, 2, True) for r in range(2)]
    assert u[0]["n_sblk"] == u[1]["n_sblk"] == _plan("ba_banded", 2, 0, 1, True)["n_sblk"]


def test_rejects_bad_input():
    lam = orc.lambda_structure(synth.make("se2_small"))[0]  # one block width: no landmark part
    with pytest.raises(api.SppError):
        api.schur_plan_host(lam)
