"""times the MFMA f64 trailing-update kernel on a Venice-sized update (m = n = 5000, k = 128)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_plus_plus_amd import api
ctx = api.Context(0, 0)
m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
k = 128
ld = 5248
A = api.DeviceArray.from_host(ctx, np.random.default_rng(0).standard_normal(ld * k if False else k * 0 + ld * 128))
P = api.DeviceArray.from_host(ctx, np.random.default_rng(1).standard_normal(128 * 0 + ld * n)[: ld * n])
C = api.DeviceArray.from_host(ctx, np.zeros(ld * n))
for it in range(3):
    ctx._check(ctx.lib.spp_dense_gemm_tn_sub(ctx.h, m, n, k, P.ptr, ld, P.ptr, ld, C.ptr, ld))
t = time.perf_counter()
reps = 20
for it in range(reps):
    ctx._check(ctx.lib.spp_dense_gemm_tn_sub(ctx.h, m, n, k, P.ptr, ld, P.ptr, ld, C.ptr, ld))
dt = (time.perf_counter() - t) / reps
print("cfg", os.environ.get("SPP_GEMM_CFG", "0"), "m=n=%d k=%d: %.3f ms, %.2f TFLOP/s (full square)" % (m, k, dt * 1e3, 2.0 * m * n * k / dt * 1e-12))
