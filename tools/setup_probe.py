"""Where the set-up of a batch goes (fresh context): Lambda structure + assembly plan, symbolic analysis, per phase
(SPP_VERBOSE laps of the library) -- run on the GPU box:   SPP_VERBOSE=1 python tools/setup_probe.py venice871"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from slam_plus_plus_amd import api, synth
name = sys.argv[1] if len(sys.argv) > 1 else "venice871"
prob = synth.make(name)
c0 = api.Context(0, 0)  # process-wide one-time set-up
st0 = c0.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
c0.analyze(st0, api.MODE_AUTO)
c0.close()
for rep in range(2):
    t0 = time.perf_counter()
    c = api.Context(0, 0)
    t1 = time.perf_counter()
    st = c.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    t2 = time.perf_counter()
    c.analyze(st, api.MODE_AUTO)
    t3 = time.perf_counter()
    print("%s rep %d: context %.1f ms, assemble_analyze %.1f ms, analyze %.1f ms" % (name, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), file=sys.stderr)
    c.close()
