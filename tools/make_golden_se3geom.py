"""Golden vectors of the reference's SE(3) pose-pose edge geometry (C3DJacobians::Absolute_to_Relative with its
forward-difference Jacobians, the CEdgePose3D error and the vertex (+)): runs oracle/_ref/dropin_driver se3dump
(reference code compiled from /root/reference, CPU only) and stores tests/golden/se3_geometry.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "g.txt")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dropin_driver"), "se3dump", str(n), path], check=True)
    rows = np.array([[float(x) for x in ln.split()[1:]] for ln in open(path) if ln.startswith("S ")])
o = np.cumsum([0, 6, 6, 6, 6, 6, 36, 36, 6, 6])
assert rows.shape == (n, o[-1])
names = ["v1", "v2", "z", "e", "err", "H1", "H2", "inc", "composed"]
out = os.path.join(ROOT, "tests", "golden", "se3_geometry.npz")
np.savez_compressed(out, **{k: rows[:, o[i]:o[i + 1]] for i, k in enumerate(names)})
print(out, rows.shape)
