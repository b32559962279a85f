import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SPP_VERBOSE"] = "1"
import numpy as np
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
for n, nl in [(300, 150), (600, 300), (1000, 600), (1500, 900), (2500, 1500), (3500, 2099)]:
    prob = synth.se2_problem(n, nl, 12)
    lam, eta = orc.assemble(prob)
    s = api.CLinearSolver_HIP(mode=api.MODE_SPARSE)
    x = eta.copy()
    ok = s.Solve_PosDef_Blocky(lam, x)
    res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta) if ok else -1
    print("n", n, "ok", ok, "res", res, flush=True)
