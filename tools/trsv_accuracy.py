"""Accuracy of spp_dense_posv on ill-conditioned SPD systems (sizes with partial last blocks, condition numbers up
to 1e12): forward error against numpy.linalg.solve and the normwise backward error, for the substitution form the
environment selects (SPP_TRSV_MFORM, SPP_TRSV_CHAIN).

    python tools/trsv_accuracy.py [n ...]
"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api

sizes = [int(a) for a in sys.argv[1:]] or [100, 180, 256, 257, 300, 640, 1000]
ctx = api.Context(0, 0)
for n in sizes:
    for cond in (1e2, 1e6, 1e10, 1e12):
        rng = np.random.default_rng(n)
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = (Q * np.logspace(0, -np.log10(cond), n)) @ Q.T
        A = 0.5 * (A + A.T)
        b = rng.standard_normal(n)
        dA = api.DeviceArray.from_host(ctx, np.asfortranarray(A).ravel(order="F"))
        db = api.DeviceArray.from_host(ctx, b)
        ctx._check(ctx.lib.spp_dense_posv(ctx.h, dA.ptr, n, n, db.ptr))
        x = db.download()
        xr = np.linalg.solve(A, b)
        print("n %5d cond %.0e  |x-xr|/|xr| %.2e  backward %.2e   (numpy backward %.2e)" % (
            n, cond, np.linalg.norm(x - xr) / np.linalg.norm(xr),
            np.linalg.norm(A @ x - b) / (np.linalg.norm(A, 2) * np.linalg.norm(x)),
            np.linalg.norm(A @ xr - b) / (np.linalg.norm(A, 2) * np.linalg.norm(xr))))

