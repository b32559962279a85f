"""Wall clock of the host-only symbolic Schur plan (spp_schur_plan_host) of a BA-shaped structure, for several thread counts.
No GPU needed.   python tools/plan_time.py [workload] [threads ...]"""
import os, sys, time, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
name = sys.argv[1] if len(sys.argv) > 1 else "venice871"
threads = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16]
if os.environ.get("PLAN_CHILD"):
    import numpy as np
    from slam_plus_plus_amd import api, synth
    t = time.perf_counter()
    prob = synth.make(name)
    from oracle import spp_oracle as orc
    lam = orc.lambda_structure(prob)[0]
    t_gen = time.perf_counter() - t
    best = None
    for rep in range(3):
        d = api.schur_plan_host(lam)
        best = d if best is None or d["seconds"] < best["seconds"] else best
    print("threads %s  plan %.1f ms  (generate %.1f s)  %s" % (os.environ.get("SPP_PLAN_THREADS"), best["seconds"] * 1e3, t_gen,
          {k: v for k, v in best.items() if k != "seconds"}))
else:
    for nt in threads:
        env = dict(os.environ, PLAN_CHILD="1", SPP_PLAN_THREADS=str(nt))
        subprocess.run([sys.executable, __file__, name], env=env, check=True)
