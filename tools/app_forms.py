"""Runs the unmodified slam_plus_plus application (oracle/_ref/slam_plus_plus_{hip,ref}, `make -C oracle apps`) on the
generated BA graph of the drop-in tests under every form of the dense backward substitution and prints the residual
norms and chi2 it reports. The last residual norm (~6e-4 after five LM iterations) sits at the noise floor of this
problem: nudging every entry of one solve's result by one ulp moves it between 0.0006 and 0.0007 (measured, round 3),
the M form (results within 5 ulp of the two-tile form in every solve) prints 0.0008.

    python tools/app_forms.py
"""
import os, re, subprocess, sys, tempfile
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import conftest
with tempfile.TemporaryDirectory() as g:
    graphs = conftest._write_app_graphs(g)
    for which, env in (("ref", {}), ("hip", {"SPP_TRSV_MFORM": "0"}), ("hip", {"SPP_TRSV_MFORM": "1"}),
                       ("hip", {"SPP_TRSV_CHAIN": "1"}), ("hip", {"SPP_TRSV_CHAIN": "0"})):
        with tempfile.TemporaryDirectory() as tmp:
            e = dict(os.environ)
            e.update(env)
            p = subprocess.run([os.path.join(root, "oracle/_ref/slam_plus_plus_" + which), "-i", graphs["ba"], "-nb", "-ns"],
                               cwd=tmp, env=e, capture_output=True, text=True, timeout=300)
            print(which, env, re.findall(r"residual norm: ([-+0-9.eE]+)", p.stdout), re.findall(r"chi2 error: ([-+0-9.eE]+)", p.stdout))
