"""Looks for the sporadic 20-80 ms stalls in a loop of short solves: runs device-resident Lambda-solves for a given number
of seconds and prints WHEN (seconds since the start of the loop) every iteration longer than 5 ms happened and how long it
took. A fixed period between the outliers points at something outside the process (a sampler touching the GPU's
management interface); outliers tied to the iteration count at something inside it.

    python tools/stall_probe.py [workload] [seconds]
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc  # (checker side: only used to build Lambda for the probe)

name = sys.argv[1] if len(sys.argv) > 1 else "sphere2500"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
prob = synth.make(name)
lam, eta = orc.assemble(prob)
ctx = api.Context(0, 0)
ctx.analyze(lam, api.MODE_AUTO)
d_vals = api.DeviceArray.from_host(ctx, lam.vals)
d_eta0 = api.DeviceArray.from_host(ctx, eta)
d_eta = api.DeviceArray(ctx, eta.size)
for _ in range(5):
    d_eta.copy_from(d_eta0)
    ctx.factor_solve_device(d_vals.ptr, d_eta.ptr)
ctx.synchronize()
t_start = time.perf_counter()
n = 0
worst = []
while time.perf_counter() - t_start < secs:
    t0 = time.perf_counter()
    d_eta.copy_from(d_eta0)
    ctx.factor_solve_device(d_vals.ptr, d_eta.ptr)   # fetches the status: a host round trip per iteration
    dt = time.perf_counter() - t0
    n += 1
    if dt > 5e-3:
        worst.append((t0 - t_start, 1e3 * dt, n))
print("%s: %d iterations in %.1f s, median-ish %.3f ms; iterations > 5 ms:" % (name, n, secs, 1e3 * secs / n))
for t, ms, i in worst:
    print("  at %7.3f s  %.1f ms  (iteration %d)" % (t, ms, i))
