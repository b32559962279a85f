import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api, synth
name = sys.argv[1]
prob = synth.make(name)
ctx = api.Context(0, api.FLAG_PROFILE)
st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
d_in = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
dv, de, dr = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n), api.DeviceArray(ctx, st.n)
ctx.assemble_device(d_in[0].ptr, d_in[1].ptr, d_in[2].ptr, d_in[3].ptr, prob.damping, dv.ptr, de.ptr)
for mode in [int(m) for m in sys.argv[2:]]:
    t0 = time.time(); ctx.analyze(st, mode); ta = time.time() - t0
    for _ in range(3):
        dr.copy_from(de); assert ctx.factor_solve_device(dv.ptr, dr.ptr) == 0
    ctx.set_profiling(False); ctx.synchronize(); t = time.perf_counter()
    for _ in range(10):
        dr.copy_from(de); ctx.factor_solve_device(dv.ptr, dr.ptr)
    ctx.synchronize(); ms = 1e2 * (time.perf_counter() - t); ctx.set_profiling(True)
    dr.copy_from(de); ctx.factor_solve_device(dv.ptr, dr.ptr)
    print(name, "mode", ctx.info("MODE"), "analyze %.2fs" % ta, "%.3f ms" % ms, "landmarks", ctx.info("N_LANDMARKS"), "levels", ctx.info("N_LEVELS"),
          {k: round(v, 3) for k, v in ctx.phase_ms().items() if v})
