"""Build recipe for libspp_hip.so (hipcc, gfx950 only) -- used by __graft_entry__.build() and by hand.

hipcc cross-compiles without a GPU; the .so is built in-tree (slam_plus_plus_amd/libspp_hip.so) so that it
travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspp_hip.so")
SOURCES = ["spp_api.cpp", "spp_symbolic.cpp", "spp_dense.hip", "spp_schur.hip", "spp_sparse.hip",
           "spp_assemble.hip", "spp_geometry.hip"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X build needs ROCm's hipcc (no CPU fallback exists)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "spp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, lib=None):
    global LIB
    if lib:
        LIB = lib
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    defs = []
    if os.path.exists(os.path.join(CSRC, "spp_sparse.hip")):
        defs.append("-DSPP_HAVE_SPARSE")
    if os.path.exists(os.path.join(CSRC, "spp_assemble.hip")):
        defs.append("-DSPP_HAVE_ASSEMBLE")
    defs += os.environ.get("SPP_EXTRA_DEFS", "").split()  # experiments: e.g. SPP_EXTRA_DEFS="-DSPP_MFMA_444=0"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + defs + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
