"""Golden vector of the reference's Gauss-Newton loop (CNonlinearSolver_Lambda::Optimize(5, 0.01) with the
reference's own CLinearSolver_UberBlock, CPU only) on a generated 2D pose graph:
    make -C oracle dropin && python tools/make_golden_gn.py 400 200
runs oracle/_ref/dropin_driver dump ... (reference code compiled from /root/reference) and stores
edges, information, the initial states the reference derives from the edges, and the optimized states
in tests/golden/se2_gn_<n>.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "se3":      # python tools/make_golden_gn.py se3 6 40
    n_rings, n_per = int(sys.argv[2]), int(sys.argv[3])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "gn.txt")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dropin_driver"), "dump3", str(n_rings), str(n_per), path], check=True)
        rows = {"E": [], "I": [], "F": []}
        for ln in open(path):
            t = ln.split()
            if t[0] in rows:
                rows[t[0]].append([float(x) for x in t[1:]])
    edges, init, final = np.array(rows["E"]), np.array(rows["I"]), np.array(rows["F"])
    out = os.path.join(ROOT, "tests", "golden", "se3_gn_%d.npz" % init.shape[0])
    np.savez_compressed(out, edges=edges, info_diag=np.array([400.0] * 3 + [10000.0] * 3), init=init, final=final,
                        max_iter=5, threshold=0.01)
    print(out, edges.shape, init.shape, "moved by", np.abs(final - init).max())
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "lm":       # python tools/make_golden_gn.py lm 12 300 5
    n_cams, n_points, n_iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "lm.txt")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dropin_driver"), "lmdump", str(n_cams), str(n_points),
                        str(n_iters), path], check=True)
        rows = {"C": [], "P": [], "O": [], "F": []}
        for ln in open(path):
            t = ln.split()
            if t[0] in rows:
                rows[t[0]].append([float(x) for x in t[1:]])
    cams = np.array(rows["C"])
    fin = rows["F"]
    out = os.path.join(ROOT, "tests", "golden", "ba_lm_%d.npz" % n_cams)
    np.savez_compressed(out, cams=cams[:, :6], intr=cams[:, 6:], points=np.array(rows["P"]), obs=np.array(rows["O"]),
                        final_cams=np.array(fin[:n_cams]), final_points=np.array(fin[n_cams:]), max_iter=n_iters, threshold=0.01)
    print(out, cams.shape, len(rows["P"]), len(rows["O"]))
    sys.exit(0)
n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_loops = int(sys.argv[2]) if len(sys.argv) > 2 else 200
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "gn.txt")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dropin_driver"), "dump", str(n_poses), str(n_loops), path], check=True)
    edges, init, final = [], [], []
    for ln in open(path):
        t = ln.split()
        if t[0] == "E":
            edges.append([float(x) for x in t[1:6]])
        elif t[0] == "I":
            init.append([float(x) for x in t[1:4]])
        elif t[0] == "F":
            final.append([float(x) for x in t[1:4]])
edges, init, final = np.array(edges), np.array(init), np.array(final)
info = np.tile(np.diag([1111.11, 1111.11, 10000.0]), (edges.shape[0], 1, 1))
out = os.path.join(ROOT, "tests", "golden", "se2_gn_%d.npz" % n_poses)
np.savez_compressed(out, edges=edges, info_diag=np.array([1111.11, 1111.11, 10000.0]), init=init, final=final,
                    max_iter=5, threshold=0.01)
print(out, edges.shape, init.shape, "moved by", np.abs(final - init).max())
