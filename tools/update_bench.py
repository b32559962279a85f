"""stand-alone rate of the bulk trailing update (upper tiles, K = 128) over the sizes one factorization meets"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slam_plus_plus_amd import api
ctx = api.Context(0, 0)
sizes = [int(a) for a in sys.argv[1:]] or [5120, 4864, 4608, 4352, 4096, 3840, 3584, 3072, 2560]
env = {k: v for k, v in os.environ.items() if k.startswith("SPP_")}
out = []
for m in sizes:
    ms = ctx.microbench_update(m, 20)
    fl = 128.0 * m * (m + 1) + 256.0 * m
    out.append("%d: %.1f us %.1f TF" % (m, ms * 1e3, fl / ms * 1e-9))
print(env, " | ".join(out))
