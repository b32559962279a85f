import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api
ctx = api.Context(0, 0)
n = 127
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n)); A = M @ M.T / n + 2 * np.eye(n)
dA = api.DeviceArray.from_host(ctx, np.asfortranarray(A).ravel(order="F"))
for it in range(3):
    ctx.lib.spp_dense_potrf_upper(ctx.h, dA.ptr, n, n)
t = time.perf_counter()
for it in range(200):
    ctx.lib.spp_dense_potrf_upper(ctx.h, dA.ptr, n, n)
print("dbg", os.environ.get("SPP_POTRF_DBG", "0"), "potrf(127) call %.1f us" % ((time.perf_counter() - t) / 200 * 1e6))
