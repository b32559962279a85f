"""Golden vectors of the reference's BA edge geometry (CBAJacobians::Project_P2C with its forward-difference
Jacobians, and the SE(3) composition behind the camera (+)): runs oracle/_ref/dropin_driver badump (reference
code compiled from /root/reference, CPU only) and stores tests/golden/ba_geometry.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "g.txt")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "dropin_driver"), "badump", str(n), path], check=True)
    rows = np.array([[float(x) for x in ln.split()[1:]] for ln in open(path) if ln.startswith("S")])
assert rows.shape == (n, 6 + 5 + 3 + 2 + 12 + 6 + 6 + 6)
o = np.cumsum([0, 6, 5, 3, 2, 12, 6, 6, 6])
out = os.path.join(ROOT, "tests", "golden", "ba_geometry.npz")
np.savez_compressed(out, cam=rows[:, o[0]:o[1]], intr=rows[:, o[1]:o[2]], X=rows[:, o[2]:o[3]], uv=rows[:, o[3]:o[4]],
                    H1=rows[:, o[4]:o[5]], H2=rows[:, o[5]:o[6]], inc=rows[:, o[6]:o[7]], composed=rows[:, o[7]:o[8]])
print(out, rows.shape)
