import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api, synth
name = sys.argv[1] if len(sys.argv) > 1 else "sphere2500"
prob = synth.make(name); pg = synth.pose_graph_states(prob)
ctx = api.Context(0)
st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
dof, nv, ne = pg["dof"], pg["poses"].shape[0], pg["v0"].size
d = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(pg[k]).ravel()) for k in ("poses", "meas", "v0", "v1")}
dOm = api.DeviceArray.from_host(ctx, prob.Om.ravel())
J0, J1, r = api.DeviceArray(ctx, dof * dof * ne), api.DeviceArray(ctx, dof * dof * ne), api.DeviceArray(ctx, dof * ne)
vals, eta = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n)
lin = ctx.se2_linearize_device if dof == 3 else ctx.se3_linearize_device
upd = ctx.se2_update_device if dof == 3 else ctx.se3_update_device
def T(f, n=20):
    f(); ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize(); return 1e3 * (time.perf_counter() - t) / n
print("linearize %.3f ms" % T(lambda: lin(ne, d["v0"].ptr, d["v1"].ptr, d["poses"].ptr, d["meas"].ptr, J0.ptr, J1.ptr, r.ptr)))
print("assemble  %.3f ms" % T(lambda: ctx.assemble_device(J0.ptr, J1.ptr, dOm.ptr, r.ptr, 0.0, vals.ptr, eta.ptr)))
ctx.analyze(st, api.MODE_AUTO)
def solve():
    ctx.assemble_device(J0.ptr, J1.ptr, dOm.ptr, r.ptr, 0.0, vals.ptr, eta.ptr)
    assert ctx.factor_solve_device(vals.ptr, eta.ptr) == 0
print("assemble+solve %.3f ms" % T(solve))
print("update(norm only) %.3f ms" % T(lambda: upd(nv, d["poses"].ptr, eta.ptr, apply=False)))
pw = api.DeviceArray(ctx, pg["poses"].size)
def it():
    pw.copy_from(d["poses"])
    lin(ne, d["v0"].ptr, d["v1"].ptr, pw.ptr, d["meas"].ptr, J0.ptr, J1.ptr, r.ptr)
    ctx.assemble_device(J0.ptr, J1.ptr, dOm.ptr, r.ptr, 0.0, vals.ptr, eta.ptr)
    assert ctx.factor_solve_device(vals.ptr, eta.ptr) == 0
    return upd(nv, pw.ptr, eta.ptr, apply=True)
print("resident iteration %.3f ms" % T(it, 10), "phases", {k: round(v, 3) for k, v in ctx.phase_ms().items()})
ctx.set_profiling(True); it(); print({k: round(v, 3) for k, v in ctx.phase_ms().items()})
# the same with the problem's own J (synth parameterization)
dj = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.r)]
def it2():
    ctx.assemble_device(dj[0].ptr, dj[1].ptr, dOm.ptr, dj[2].ptr, 0.0, vals.ptr, eta.ptr)
    assert ctx.factor_solve_device(vals.ptr, eta.ptr) == 0
ctx.set_profiling(False)
print("synth-J assemble+solve %.3f ms" % T(it2, 10))
