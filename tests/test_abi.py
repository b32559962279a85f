"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/spp_hip.h declares (no compute is called: there is no GPU in the build container)."""
import ctypes
import os
import re

from slam_plus_plus_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "spp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spp_[a-z0-9_]+)\s*\(", src)) - {"spp_ctx"})


def test_library_is_built_and_loads():
    build.build()
    assert os.path.exists(api.LIB_PATH)
    ctypes.CDLL(api.LIB_PATH)


def test_every_declared_symbol_is_exported():
    build.build()
    lib = ctypes.CDLL(api.LIB_PATH)
    decl = _declared_symbols()
    assert len(decl) >= 25
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, "declared in include/spp_hip.h but not exported: %s" % missing
    assert sorted(api.EXPORTS) == decl, "api.EXPORTS out of sync with the header"


def test_no_cpu_fallback_create_fails_without_gpu():
    """On a machine without a HIP device spp_create returns NULL and the binding raises: the
    product path never silently computes on the CPU."""
    import pytest
    lib = api.load_library()
    h = lib.spp_create(0, 0)
    if h:  # running on a GPU box: nothing to check here
        lib.spp_destroy(h)
        pytest.skip("a HIP device is present")
    with pytest.raises(api.SppError):
        api.Context(0)
    with pytest.raises(api.SppError):
        api.CLinearSolver_HIP()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "slam_plus_plus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "spp_oracle" not in text and "libspp_ref" not in text, f
