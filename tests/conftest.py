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


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    if not os.path.exists("/dev/kfd"):
        return
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for key, name, args in (("dropin_driver", "dropin_driver", ["400", "200"]),
                            ("dropin_driver_ba", "dropin_driver", ["ba", "16", "600", "1"]),
                            ("dropin_driver_ba3", "dropin_driver", ["ba", "40", "3000", "3"]),
                            ("slam_simple_hip", "slam_simple_hip", [])):
        exe = os.path.join(ref_dir, name)
        if not os.path.exists(exe):
            continue
        with tempfile.TemporaryDirectory() as tmp:
            try:
                p = subprocess.run([exe] + args, cwd=tmp, env=env, capture_output=True, text=True, timeout=300)
                DROPIN_RESULTS[key] = (p.returncode, p.stdout, p.stderr, sorted(os.listdir(tmp)))
            except Exception as e:  # noqa: BLE001
                DROPIN_RESULTS[key] = (-999, "", repr(e), [])
