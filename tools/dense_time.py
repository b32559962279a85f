"""Times the dense factor alone (spp_dense_potrf_upper, rhs column carried along) on a reduced-camera-system
sized SPD matrix, and the vendor path (torch.linalg.cholesky: hipSOLVER / MAGMA) on the same matrix.

    python tools/dense_time.py [n] [reps] [--vendor] [--solve]

--solve: factor + backward substitution (spp_dense_posv) timed as well; the difference to the factor alone is the
substitution (SPP_TRSV_CHAIN / SPP_TRSV2_CFG select its form).
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if "--vendor" in sys.argv:
    import torch
    torch.cuda.init()
from slam_plus_plus_amd import api

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 5226
reps = int(args[1]) if len(args) > 1 else 20
ld = (n + 1 + 127) // 128 * 128
rng = np.random.default_rng(0)
M = rng.standard_normal((n, n))
A = M @ M.T / n + 2.0 * np.eye(n)
Ap = np.eye(ld)
Ap[:n, :n] = A
Ap[:n, n] = rng.standard_normal(n)
ctx = api.Context(0, 0)
d0 = api.DeviceArray.from_host(ctx, np.asfortranarray(Ap).ravel(order="F"))
d1 = api.DeviceArray(ctx, ld * ld)


def run(k, factor=True):
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        d1.copy_from(d0)
        if factor:
            ctx._check(ctx.lib.spp_dense_potrf_upper(ctx.h, d1.ptr, n, ld))
    ctx.synchronize()
    return (time.perf_counter() - t) / k * 1e3


run(3)
t_copy = run(reps, False)
t_all = run(reps)
R = d1.download().reshape(ld, ld, order="F")[:n, :n]
R = np.triu(R)
err = np.linalg.norm(R.T @ R - A) / np.linalg.norm(A)
ms = t_all - t_copy
if "--solve" in sys.argv:
    b = rng.standard_normal(n)
    dA = api.DeviceArray.from_host(ctx, np.asfortranarray(A).ravel(order="F"))
    dA1 = api.DeviceArray(ctx, n * n)
    db0 = api.DeviceArray.from_host(ctx, b)
    db = api.DeviceArray(ctx, n)

    def run_posv(k, solve):
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            dA1.copy_from(dA)
            db.copy_from(db0)
            if solve:
                ctx._check(ctx.lib.spp_dense_posv(ctx.h, dA1.ptr, n, n, db.ptr))
            else:
                ctx._check(ctx.lib.spp_dense_potrf_upper(ctx.h, dA1.ptr, n, n))
        ctx.synchronize()
        return (time.perf_counter() - t) / k * 1e3

    run_posv(3, True)
    t_f = run_posv(reps, False)
    t_s = run_posv(reps, True)
    x = db.download()
    xr = np.linalg.solve(A, b)
    print("posv %.3f ms, potrf (unpadded input) %.3f ms: substitution %.3f ms  |x - x_ref|/|x_ref| %.2e" % (
        t_s, t_f, t_s - t_f, np.linalg.norm(x - xr) / np.linalg.norm(xr)))
print("n %d ld %d  factor %.3f ms (copy %.3f ms)  %.2f TFLOP/s  ||R^T R - A||/||A|| %.2e  env %s" % (
    n, ld, ms, t_copy, n ** 3 / 3.0 / ms / 1e9, err,
    {k: v for k, v in os.environ.items() if k.startswith("SPP_")}))

if "--vendor" in sys.argv:
    tA = torch.from_numpy(A).cuda()
    for _ in range(3):
        torch.linalg.cholesky(tA, upper=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        torch.linalg.cholesky(tA, upper=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print("vendor torch.linalg.cholesky (%s) %.3f ms  %.2f TFLOP/s" % (
        torch.backends.cuda.preferred_linalg_library(), ms, n ** 3 / 3.0 / ms / 1e9))
