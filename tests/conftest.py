import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One spp context for the whole GPU session. Fails loudly if the HIP library is missing."""
    from slam_plus_plus_amd import api
    ctx = api.Context(0, api.FLAG_PROFILE)
    yield ctx
    ctx.close()


# ---- drop-in binaries (reference nonlinear solver + our linear solver), built by `make -C oracle dropin`
# into oracle/_ref/. They are separate programs: run them BEFORE this process initialises HIP (a
# process that already owns the GPU must not exec another program on the GPU boxes) and cache what
# they printed for tests/test_gpu_dropin.py.
DROPIN_RESULTS = {}
DROPIN_FILES = {}  # key -> {file name: text} for the small text files a program left behind (solution.txt)


def _write_app_graphs(tmp):
    """graph files in the reference's text formats for its own applications (numpy only: no HIP here)"""
    import numpy as np
    from slam_plus_plus_amd import synth, formats
    out = {}
    p = synth.make("se2_small")
    st = synth.pose_graph_states(p)
    e = np.concatenate([st["v0"][:, None].astype(float), st["v1"][:, None].astype(float), st["meas"]], axis=1)
    out["se2"] = os.path.join(tmp, "se2_small.txt")
    formats.save_se2_graph(out["se2"], st["poses"], e, p.Om.reshape(-1, 3, 3))
    # 3D poses on a helix with loop closures; relative poses [R_i^T (t_j - t_i) | log(R_i^T R_j)] + small noise, edges
    # only (the reference composes the initial poses from them, as it does for sphere2500): Gauss-Newton converges
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    n = 80
    k = np.arange(n)
    t = np.stack([8 * np.cos(0.25 * k), 8 * np.sin(0.25 * k), 0.15 * k], axis=1)
    R = Rotation.from_euler("zyx", np.stack([0.25 * k + np.pi / 2, 0.05 * np.sin(0.3 * k), 0.03 * np.cos(0.2 * k)], axis=1))
    pairs = [(i, i + 1) for i in range(n - 1)] + [(i, i + 25) for i in range(0, n - 25, 3)] + \
            [(i, i + 50) for i in range(0, n - 50, 7)]
    e = []
    for a, b in pairs:
        zt = R[a].inv().apply(t[b] - t[a]) + rng.normal(0, 0.02, 3)
        zr = (R[a].inv() * R[b] * Rotation.from_rotvec(rng.normal(0, 0.004, 3))).as_rotvec()
        e.append([a, b, *zt, *zr])
    out["se3"] = os.path.join(tmp, "se3_app.txt")
    formats.save_se3_graph(out["se3"], np.array(e), np.tile(np.diag([400.0] * 3 + [1e4] * 3), (len(e), 1, 1)))
    p = synth.make("ba_small")
    st = synth.ba_states(p)
    o = np.stack([st["pt_of"].astype(float), st["cam_of"].astype(float), st["meas"][:, 0], st["meas"][:, 1]], axis=1)
    out["ba"] = os.path.join(tmp, "ba_small.txt")
    formats.save_ba_graph(out["ba"], st["cams"], st["intr"], st["points"], o)
    # BASELINE config 3 shape (Ladybug-49: 49 cameras, 7 776 points, 31 843 observations) for the applications
    p = synth.make("ladybug49")
    st = synth.ba_states(p)
    o = np.stack([st["pt_of"].astype(float), st["cam_of"].astype(float), st["meas"][:, 0], st["meas"][:, 1]], axis=1)
    out["ladybug"] = os.path.join(tmp, "ladybug49.txt")
    formats.save_ba_graph(out["ladybug"], st["cams"], st["intr"], st["points"], o)
    return out


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    if not os.path.exists("/dev/kfd"):
        return
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    env = dict(os.environ, OMP_NUM_THREADS="1")

    def run(key, name, args, tmp, extra_env=None):
        exe = os.path.join(ref_dir, name)
        if not os.path.exists(exe):
            return
        try:
            p = subprocess.run([exe] + args, cwd=tmp, env=dict(env, **(extra_env or {})), capture_output=True, text=True, timeout=300)
            DROPIN_RESULTS[key] = (p.returncode, p.stdout, p.stderr, sorted(os.listdir(tmp)))
            sol = os.path.join(tmp, "solution.txt")
            if os.path.exists(sol) and os.path.getsize(sol) < (8 << 20):
                DROPIN_FILES[key] = {"solution.txt": open(sol).read()}
        except Exception as e:  # noqa: BLE001
            DROPIN_RESULTS[key] = (-999, "", repr(e), [])

    for key, name, args in (("dropin_driver", "dropin_driver", ["400", "200"]),
                            ("dropin_driver_ba", "dropin_driver", ["ba", "16", "600", "1"]),
                            ("dropin_driver_ba3", "dropin_driver", ["ba", "40", "3000", "3"]),
                            ("slam_simple_hip", "slam_simple_hip", [])):
        with tempfile.TemporaryDirectory() as tmp:
            run(key, name, args, tmp)
    # the adapter on a Venice-sized CUberBlockMatrix (871 poses, 530 304 landmarks, 2.65 M pose-landmark blocks, 420 MB):
    # Flatten_Values with up to 16 host threads + the host-pointer solve, timed by the adapter itself
    with tempfile.TemporaryDirectory() as tmp:
        run("dropin_adapter_scale", "dropin_driver", ["adapter", "871", "530304", "5"], tmp)
    # the adapter's threaded flatten (taken by itself from 32 MB of Lambda on) forced onto a small system
    with tempfile.TemporaryDirectory() as tmp:
        run("dropin_driver_ba_mt", "dropin_driver", ["ba", "16", "600", "1"], tmp, {"SPP_ADAPTER_FLATTEN_THREADS": "5"})
    # the reference's applications, sources untouched: slam_plus_plus (src/slam_app) and ba_iface_example, each
    # built twice by `make -C oracle apps` -- *_hip with the shim directory first on the include path, *_ref
    # without it -- on the same generated graph files (scripts/tests/unit_tests.sh style: iterations + chi2)
    if os.path.exists(os.path.join(ref_dir, "slam_plus_plus_hip")):
        with tempfile.TemporaryDirectory() as gdir:
            graphs = _write_app_graphs(gdir)
            for kind, extra in (("se2", ["-po"]), ("se3", ["-po"]), ("ba", []), ("ba_us", ["-us"]), ("ladybug", [])):
                g = graphs[kind.split("_")[0]]
                for which in ("hip", "ref"):
                    with tempfile.TemporaryDirectory() as tmp:
                        run("app_%s_%s" % (kind, which), "slam_plus_plus_" + which, ["-i", g, "-nb", "-ns"] + extra, tmp)
            for which in ("hip", "ref"):
                with tempfile.TemporaryDirectory() as tmp:
                    run("bax_%s" % which, "ba_iface_" + which, ["-i", graphs["ba"], "-q"], tmp)
