import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPP_VERBOSE"] = "1"
from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc
for name in sys.argv[1:]:
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    ctx = api.Context(0)
    ctx.analyze(lam, api.MODE_SPARSE)
    if orc.have_ref():
        import numpy as np, ctypes
        rs = orc.RefSolver("uberblock", lam)
        a, b = ctypes.c_int64(), ctypes.c_int64()
        orc.ref().ref_factor_fill(rs.h, ctypes.byref(a), ctypes.byref(b))
        print(name, "reference AMD fill: R blocks", a.value, "scalar nnz", b.value, file=sys.stderr)
