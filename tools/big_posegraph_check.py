"""Scale check of the sparse path's dependency-driven launch: 2D pose graphs of 20 000 and 60 000 poses (5 000 / 16 000
supernodes -- many times the workgroups a launch holds resident), residual, bit-reproducibility and time per solve.

    python tools/big_posegraph_check.py        (on the GPU box; uses the oracle only to assemble Lambda)
"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
for (n, loops) in ((20000, 12000), (60000, 36000)):
    prob = synth.se2_problem(n, loops, 77, name="se2_big")
    lam, eta = orc.assemble(prob)
    s = api.CLinearSolver_HIP(mode=api.MODE_SPARSE)
    x = eta.copy()
    t = time.time(); ok = s.Solve_PosDef_Blocky(lam, x); t1 = time.time() - t
    res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta)
    xs = []
    for _ in range(3):
        x2 = eta.copy(); t = time.time(); s.Solve_PosDef_Blocky(lam, x2); xs.append(time.time() - t)
    print("n_poses", n, "ok", ok, "residual %.2e" % res, "supernodes", s.ctx.info("N_SUPERNODES"), "levels", s.ctx.info("N_LEVELS"),
          "first %.1f ms (with analysis), then %.2f ms per host-pointer solve" % (1e3 * t1, 1e3 * min(xs)), "bitrepro", np.array_equal(x, x2))
