"""ctypes binding of libspp_hip.so (include/spp_hip.h) plus a host-side mirror of the reference's
linear-solver concept.

`CLinearSolver_HIP` mirrors the duck-typed interface every SLAM++ linear solver implements
(reference include/slam/LinearSolverTags.h:38-135, include/slam/LinearSolver_UberBlock.h:44-427):
same method names, same argument meaning (Lambda upper block triangle + eta overwritten by the
solution), same error behaviour (False = not positive definite, exceptions for everything else).
The C++ twin for linking into the reference itself is include/spp_adapter.h.

There is no CPU fallback here: if the HIP library or a GPU is missing, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPP_LIB") or os.path.join(_HERE, "libspp_hip.so")  # SPP_LIB: another build of the library (A/B experiments)

SPP_OK, SPP_NOT_POSDEF = 0, 1
MODE_AUTO, MODE_SPARSE, MODE_SCHUR, MODE_SCHUR_SPARSE, MODE_SCHUR_MIS = 0, 1, 2, 3, 4
FLAG_PROFILE = 1
INFO = dict(MODE=0, N=1, NNZB=2, NVALS=3, FACTOR_NNZ=4, FACTOR_FLOPS=5, N_REDUCED=6, N_POSES=7,
            N_LANDMARKS=8, SCHUR_PAIRS=9, N_OBS=10, SOLVE_BYTES=11, N_SUPERNODES=12, N_LEVELS=13, S_LD=14, S_NNZB=15, DENSE_STREAMED=16)
PHASES = ["permute", "schur_inv", "schur_gemm", "schur_rhs", "factor", "trisolve", "backsubst", "assemble", "total"]

# every symbol include/spp_hip.h declares (tests/test_abi.py checks the .so exports them all)
EXPORTS = [
    "spp_create", "spp_destroy", "spp_free_memory", "spp_last_error", "spp_host_staging", "spp_set_stream", "spp_synchronize",
    "spp_analyze", "spp_set_shard", "spp_get_info", "spp_get_ordering", "spp_factor_solve",
    "spp_factor_solve_device", "spp_schur_buffer_size", "spp_schur_form", "spp_schur_finish",
    "spp_schur_packed_size", "spp_schur_pack", "spp_schur_unpack",
    "spp_assemble_analyze", "spp_assemble_get_structure", "spp_assemble_device", "spp_assemble_set_edge_weights", "spp_device_malloc",
    "spp_device_free", "spp_memcpy_h2d", "spp_memcpy_d2h", "spp_memcpy_d2d", "spp_get_phase_ms", "spp_get_dominant_kernel",
    "spp_microbench_copy", "spp_microbench_mfma_f64", "spp_microbench_ctile", "spp_microbench_update", "spp_block_ordering", "spp_schur_plan_host", "spp_set_profiling", "spp_se2_linearize_device", "spp_se2_update_device", "spp_ba_linearize_device", "spp_ba_update_device", "spp_se3_linearize_device", "spp_se3_update_device", "spp_edge_chi2_device", "spp_edge_robust_weights_device", "spp_edge_hessian_maxdiag_device",
    "spp_lm_gain_denominator_device", "spp_dense_potrf_upper", "spp_dense_posv",
    "spp_dense_gemm_tn_sub", "spp_version",
]

_lib = None
_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f64p = ctypes.POINTER(ctypes.c_double)


class SppError(RuntimeError):
    """std::runtime_error of the C++ adapter (reference LinearSolver_Schur_GPU.cpp:734-797)."""


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SppError("libspp_hip.so is not built: run `python -m slam_plus_plus_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, dbl, cint = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, ctypes.c_int
    sig = {
        "spp_create": (vp, [cint, cint]),
        "spp_destroy": (None, [vp]),
        "spp_free_memory": (cint, [vp]),
        "spp_last_error": (cint, [vp, ctypes.c_char_p, ctypes.c_size_t]),
        "spp_host_staging": (vp, [vp, i64]),
        "spp_set_stream": (cint, [vp, vp]),
        "spp_synchronize": (cint, [vp]),
        "spp_analyze": (cint, [vp, i64, vp, vp, vp, vp, cint]),
        "spp_set_shard": (cint, [vp, cint, cint]),
        "spp_get_info": (cint, [vp, cint, _c_i64p]),
        "spp_get_ordering": (cint, [vp, vp]),
        "spp_factor_solve": (cint, [vp, vp, vp]),
        "spp_factor_solve_device": (cint, [vp, vp, vp]),
        "spp_schur_buffer_size": (cint, [vp, _c_i64p]),
        "spp_schur_form": (cint, [vp, vp, vp, vp]),
        "spp_schur_finish": (cint, [vp, vp, vp, vp]),
        "spp_schur_packed_size": (cint, [vp, _c_i64p]),
        "spp_schur_pack": (cint, [vp, vp, vp]),
        "spp_schur_unpack": (cint, [vp, vp, vp]),
        "spp_assemble_analyze": (cint, [vp, i64, vp, i64, vp, vp, cint, cint, cint, i64]),
        "spp_assemble_get_structure": (cint, [vp, vp, vp, vp]),
        "spp_assemble_device": (cint, [vp, vp, vp, vp, vp, dbl, vp, vp]),
        "spp_assemble_set_edge_weights": (cint, [vp, vp]),
        "spp_device_malloc": (cint, [vp, ctypes.c_size_t, ctypes.POINTER(vp)]),
        "spp_device_free": (cint, [vp, vp]),
        "spp_memcpy_h2d": (cint, [vp, vp, vp, ctypes.c_size_t]),
        "spp_memcpy_d2h": (cint, [vp, vp, vp, ctypes.c_size_t]),
        "spp_memcpy_d2d": (cint, [vp, vp, vp, ctypes.c_size_t]),
        "spp_get_phase_ms": (cint, [vp, _c_f64p]),
        "spp_get_dominant_kernel": (cint, [vp, _c_f64p, _c_i64p, _c_f64p]),
        "spp_microbench_copy": (cint, [vp, ctypes.c_size_t, cint, _c_f64p]),
        "spp_microbench_mfma_f64": (cint, [vp, cint, _c_f64p]),
        "spp_microbench_ctile": (cint, [vp, cint, cint, _c_f64p]),
        "spp_microbench_update": (cint, [vp, ctypes.c_int64, cint, _c_f64p]),
        "spp_block_ordering": (cint, [ctypes.c_int64, vp, vp, cint, vp]),
        "spp_schur_plan_host": (cint, [ctypes.c_int64, vp, vp, vp, cint, cint, cint, vp, vp]),
        "spp_set_profiling": (cint, [vp, cint]),
        "spp_se2_linearize_device": (cint, [vp, ctypes.c_int64, vp, vp, vp, vp, vp, vp, vp]),
        "spp_se2_update_device": (cint, [vp, ctypes.c_int64, vp, vp, cint, _c_f64p]),
        "spp_se3_linearize_device": (cint, [vp, ctypes.c_int64, vp, vp, vp, vp, vp, vp, vp]),
        "spp_se3_update_device": (cint, [vp, ctypes.c_int64, vp, vp, cint, _c_f64p]),
        "spp_edge_chi2_device": (cint, [vp, ctypes.c_int64, cint, vp, vp, _c_f64p]),
        "spp_edge_robust_weights_device": (cint, [vp, ctypes.c_int64, cint, cint, dbl, dbl, vp, vp]),
        "spp_edge_hessian_maxdiag_device": (cint, [vp, ctypes.c_int64, cint, cint, cint, vp, vp, vp, _c_f64p]),
        "spp_lm_gain_denominator_device": (cint, [vp, ctypes.c_int64, vp, vp, ctypes.c_double, _c_f64p]),
        "spp_ba_linearize_device": (cint, [vp, ctypes.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "spp_ba_update_device": (cint, [vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp, vp, vp, ctypes.c_int64, cint, _c_f64p]),
        "spp_dense_potrf_upper": (cint, [vp, vp, i64, i64]),
        "spp_dense_posv": (cint, [vp, vp, i64, i64, vp]),
        "spp_dense_gemm_tn_sub": (cint, [vp, i64, i64, i64, vp, i64, vp, i64, vp, i64]),
        "spp_version": (ctypes.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


ORDER_AMD, ORDER_ND = 0, 1


def block_ordering(lam, method=ORDER_AMD):
    """Host-only fill-reducing ordering of a BlockCSC / Structure pattern (spp_block_ordering): the
    counterpart of CMatrixOrdering::p_BlockOrdering, src/slam/OrderingMagic.cpp:701. No GPU needed."""
    lib = load_library()
    col_ptr = np.ascontiguousarray(lam.col_ptr, dtype=np.int64)
    row_idx = np.ascontiguousarray(lam.row_idx, dtype=np.int64)
    out = np.empty(lam.nb, dtype=np.int64)
    code = lib.spp_block_ordering(lam.nb, _ptr(col_ptr), _ptr(row_idx), method, _ptr(out))
    if code != 0:
        raise SppError("spp_block_ordering failed: %d" % code)
    return out


def schur_plan_host(lam, shard_rank=0, shard_world=1, sparse_S=False):
    """Host-only symbolic Schur plan of a BlockCSC pattern (spp_schur_plan_host): returns a dict with the list sizes,
    a checksum of the lists and the wall clock of the plan. No GPU needed."""
    lib = load_library()
    col_ptr = np.ascontiguousarray(lam.col_ptr, dtype=np.int64)
    row_idx = np.ascontiguousarray(lam.row_idx, dtype=np.int64)
    dim = np.ascontiguousarray(lam.dim, dtype=np.int32)
    out = np.zeros(8, dtype=np.int64)
    sec = ctypes.c_double(0.0)
    code = lib.spp_schur_plan_host(lam.nb, _ptr(dim), _ptr(col_ptr), _ptr(row_idx), shard_rank, shard_world,
                                   1 if sparse_S else 0, _ptr(out), ctypes.byref(sec))
    if code != 0:
        raise SppError("spp_schur_plan_host failed: %d" % code)
    keys = ("nc", "nl", "no", "n_pairs", "n_sblk", "n_items", "n_multi", "checksum")
    d = dict(zip(keys, (int(v) for v in out)))
    d["seconds"] = sec.value
    return d


class DeviceArray:
    """A raw HBM allocation owned through the C ABI (no torch involved)."""

    def __init__(self, ctx, n, dtype=np.float64):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        p = ctypes.c_void_p()
        ctx._check(ctx.lib.spp_device_malloc(ctx.h, self.n * self.dtype.itemsize, ctypes.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        d = cls(ctx, arr.size, arr.dtype)
        d.upload(arr)
        return d

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.n
        self.ctx._check(self.ctx.lib.spp_memcpy_h2d(self.ctx.h, self.ptr, _ptr(arr), arr.nbytes))

    def download(self):
        out = np.empty(self.n, dtype=self.dtype)
        self.ctx._check(self.ctx.lib.spp_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out

    def copy_from(self, other):
        """asynchronous device-to-device copy on the ctx stream"""
        assert other.n * other.dtype.itemsize <= self.n * self.dtype.itemsize
        self.ctx._check(self.ctx.lib.spp_memcpy_d2d(self.ctx.h, self.ptr, other.ptr, other.n * other.dtype.itemsize))

    def free(self):
        if self.ptr:
            self.ctx.lib.spp_device_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """Thin object wrapper over spp_ctx."""

    def __init__(self, device=0, flags=0):
        self.lib = load_library()
        self.h = self.lib.spp_create(int(device), int(flags))
        if not self.h:
            raise SppError("spp_create failed: no usable HIP device %d (the solver has no CPU fallback)" % device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.spp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        buf = ctypes.create_string_buffer(1024)
        self.lib.spp_last_error(self.h, buf, 1024)
        return buf.value.decode(errors="replace")

    def _check(self, code):
        if code < 0:
            if code == -2:
                raise MemoryError(self.last_error())
            raise SppError("spp error %d: %s" % (code, self.last_error()))
        return code

    # --- symbolic
    def analyze(self, bcsc, mode=MODE_AUTO):
        self._keep = bcsc
        return self._check(self.lib.spp_analyze(self.h, bcsc.nb, _ptr(bcsc.col_ptr), _ptr(bcsc.row_idx),
                                                 _ptr(bcsc.blk_off), _ptr(bcsc.dim), mode))

    def set_shard(self, rank, world):
        return self._check(self.lib.spp_set_shard(self.h, rank, world))

    def info(self, key):
        out = ctypes.c_int64()
        self._check(self.lib.spp_get_info(self.h, INFO[key], ctypes.byref(out)))
        return out.value

    def ordering(self, nb):
        out = np.empty(nb, dtype=np.int64)
        self._check(self.lib.spp_get_ordering(self.h, _ptr(out)))
        return out

    def se2_linearize_device(self, n_edges, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r):
        return self._check(self.lib.spp_se2_linearize_device(self.h, n_edges, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r))

    def se2_update_device(self, n_vertices, d_poses, d_dx, apply=True):
        """returns ||dx|| (host), applies x <- x (+) dx on the device when `apply`"""
        out = ctypes.c_double()
        self._check(self.lib.spp_se2_update_device(self.h, n_vertices, d_poses, d_dx, 1 if apply else 0, ctypes.byref(out)))
        return out.value ** 0.5

    def se3_linearize_device(self, n_edges, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r):
        return self._check(self.lib.spp_se3_linearize_device(self.h, n_edges, d_v0, d_v1, d_poses, d_meas, d_J0, d_J1, d_r))

    def se3_update_device(self, n_vertices, d_poses, d_dx, apply=True):
        out = ctypes.c_double()
        self._check(self.lib.spp_se3_update_device(self.h, n_vertices, d_poses, d_dx, 1 if apply else 0, ctypes.byref(out)))
        return out.value ** 0.5

    def edge_robust_weights_device(self, n_edges, rd, d_r, d_w, scale, param=1.345, kind=0):
        """w_e = Huber(||r_e|| / scale) on the device (the reference's CRobustify_ErrorNorm_Default with CHuberLossd)"""
        return self._check(self.lib.spp_edge_robust_weights_device(self.h, n_edges, rd, kind, float(scale), float(param), d_r, d_w))

    def edge_chi2_device(self, n_edges, rd, d_r, d_Om):
        out = ctypes.c_double()
        self._check(self.lib.spp_edge_chi2_device(self.h, n_edges, rd, d_r, d_Om, ctypes.byref(out)))
        return out.value

    def edge_hessian_maxdiag_device(self, n_edges, rd, d0, d1, d_J0, d_J1, d_Om):
        out = ctypes.c_double()
        self._check(self.lib.spp_edge_hessian_maxdiag_device(self.h, n_edges, rd, d0, d1, d_J0, d_J1, d_Om, ctypes.byref(out)))
        return out.value

    def lm_gain_denominator_device(self, n, d_dx, d_eta, alpha):
        out = ctypes.c_double()
        self._check(self.lib.spp_lm_gain_denominator_device(self.h, n, d_dx, d_eta, float(alpha), ctypes.byref(out)))
        return out.value

    def ba_linearize_device(self, n_obs, d_cam_of, d_pt_of, d_cams, d_intr, d_points, d_meas, d_J0, d_J1, d_r):
        return self._check(self.lib.spp_ba_linearize_device(self.h, n_obs, d_cam_of, d_pt_of, d_cams, d_intr, d_points,
                                                            d_meas, d_J0, d_J1, d_r))

    def ba_update_device(self, n_cams, d_cams, d_cam_dxoff, n_points, d_points, d_pt_dxoff, d_dx, n_dx, apply=True):
        out = ctypes.c_double()
        self._check(self.lib.spp_ba_update_device(self.h, n_cams, d_cams, d_cam_dxoff, n_points, d_points, d_pt_dxoff,
                                                  d_dx, n_dx, 1 if apply else 0, ctypes.byref(out)))
        return out.value ** 0.5

    def set_profiling(self, on):
        return self._check(self.lib.spp_set_profiling(self.h, 1 if on else 0))

    def set_stream(self, stream_handle):
        return self._check(self.lib.spp_set_stream(self.h, ctypes.c_void_p(stream_handle)))

    def synchronize(self):
        return self._check(self.lib.spp_synchronize(self.h))

    def host_staging(self, n):
        """numpy view of the ctx's page-locked host buffer (spp_host_staging): fill it with the block values and pass
        it to factor_solve() -- the host-pointer entry then copies at the DMA rate of the link"""
        p = self.lib.spp_host_staging(self.h, int(n))
        if not p:
            raise MemoryError(self.last_error())
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(int(n),))

    # --- numeric
    def factor_solve(self, vals, rhs):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        x = np.array(rhs, dtype=np.float64, copy=True)
        code = self._check(self.lib.spp_factor_solve(self.h, _ptr(vals), _ptr(x)))
        return code, x

    def factor_solve_device(self, d_vals, d_rhs):
        return self._check(self.lib.spp_factor_solve_device(self.h, d_vals, d_rhs))

    def schur_buffer_size(self):
        out = ctypes.c_int64()
        self._check(self.lib.spp_schur_buffer_size(self.h, ctypes.byref(out)))
        return out.value

    def schur_packed_size(self):
        out = ctypes.c_int64()
        self._check(self.lib.spp_schur_packed_size(self.h, ctypes.byref(out)))
        return out.value

    def schur_pack(self, d_S, d_packed):
        return self._check(self.lib.spp_schur_pack(self.h, d_S, d_packed))

    def schur_unpack(self, d_packed, d_S):
        return self._check(self.lib.spp_schur_unpack(self.h, d_packed, d_S))

    def schur_form(self, d_vals, d_rhs, d_S):
        return self._check(self.lib.spp_schur_form(self.h, d_vals, d_rhs, d_S))

    def schur_finish(self, d_vals, d_S, d_rhs):
        return self._check(self.lib.spp_schur_finish(self.h, d_vals, d_S, d_rhs))

    # --- assembly
    def assemble_analyze(self, dim, v0, v1, d0, d1, rd, unary_vertex=-1):
        dim = np.ascontiguousarray(dim, dtype=np.int32)
        v0 = np.ascontiguousarray(v0, dtype=np.int64)
        v1 = np.ascontiguousarray(v1, dtype=np.int64)
        self._check(self.lib.spp_assemble_analyze(self.h, dim.size, _ptr(dim), v0.size, _ptr(v0), _ptr(v1),
                                                   d0, d1, rd, int(unary_vertex)))
        nb, nnzb = dim.size, self.info("NNZB")
        col_ptr = np.empty(nb + 1, dtype=np.int64)
        row_idx = np.empty(nnzb, dtype=np.int64)
        blk_off = np.empty(nnzb, dtype=np.int64)
        self._check(self.lib.spp_assemble_get_structure(self.h, _ptr(col_ptr), _ptr(row_idx), _ptr(blk_off)))
        from .blockcsc import BlockCSC
        return BlockCSC(dim, col_ptr, row_idx, blk_off, None, nvals=self.info("NVALS"))

    def assemble_set_edge_weights(self, d_w):
        """robust edges: device array of one weight per edge for the following assemble_device calls (None: plain edges)"""
        return self._check(self.lib.spp_assemble_set_edge_weights(self.h, d_w))

    def assemble_device(self, d_J0, d_J1, d_Om, d_r, damping, d_vals, d_eta):
        return self._check(self.lib.spp_assemble_device(self.h, d_J0, d_J1, d_Om, d_r, float(damping), d_vals, d_eta))

    # --- profiling
    def phase_ms(self):
        out = np.zeros(len(PHASES))
        self._check(self.lib.spp_get_phase_ms(self.h, out.ctypes.data_as(_c_f64p)))
        return dict(zip(PHASES, out.tolist()))

    def dominant_kernel(self):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._check(self.lib.spp_get_dominant_kernel(self.h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value

    def microbench_copy(self, nbytes=1 << 30, iters=10):
        out = ctypes.c_double()
        self._check(self.lib.spp_microbench_copy(self.h, nbytes, iters, ctypes.byref(out)))
        return out.value

    def microbench_ctile(self, n=8192, iters=10):
        out = ctypes.c_double()
        self._check(self.lib.spp_microbench_ctile(self.h, n, iters, ctypes.byref(out)))
        return out.value

    def microbench_update(self, m=5120, iters=20):
        """ms per launch of the stand-alone bulk trailing update of an m x (m + 1) trailing matrix"""
        out = ctypes.c_double()
        self._check(self.lib.spp_microbench_update(self.h, m, iters, ctypes.byref(out)))
        return out.value

    def microbench_mfma_f64(self, iters=4000):
        out = ctypes.c_double()
        self._check(self.lib.spp_microbench_mfma_f64(self.h, iters, ctypes.byref(out)))
        return out.value


class CLinearSolver_HIP:
    """Host-side mirror of the reference's blockwise linear-solver concept.

    reference: typedef CBlockwiseLinearSolverTag _Tag (LinearSolverTags.h:54); the calling sequence
    of CNonlinearSolver_Lambda::Optimize is Clear_SymbolicDecomposition() once, then
    Solve_PosDef_Blocky(lambda, eta) per iteration (NonlinearSolver_Lambda.h:605-626).
    `r_lambda` is a blockcsc.BlockCSC (upper block triangle, values attached), `r_eta` a numpy
    vector that is overwritten with the solution on success and left untouched on failure.
    """
    _Tag = "CBlockwiseLinearSolverTag"

    def __init__(self, device=0, mode=MODE_AUTO, flags=0):
        self._ctx = Context(device, flags)
        self._mode = mode
        self._have_symbolic = False
        self._sig = None

    def Free_Memory(self):
        self._ctx._check(self._ctx.lib.spp_free_memory(self._ctx.h))
        self._have_symbolic = False

    def Clear_SymbolicDecomposition(self):
        self._have_symbolic = False

    def SymbolicDecomposition_Blocky(self, r_lambda):
        self._ctx.analyze(r_lambda, self._mode)
        self._have_symbolic = True
        self._sig = self._signature(r_lambda)
        return True

    @staticmethod
    def _signature(r_lambda):
        """Fingerprint of the block STRUCTURE (not only of its two counts: one edge replaced by another keeps both):
        a 64-bit hash of the column pointers, block rows and block sizes, recomputed on every call (xxh3 runs at memory
        speed: ~5 ms for the 3.4 M blocks of the Venice shape, microseconds for a pose graph). No shortcut through the
        arrays' identities: an array edited in place, or an id recycled after garbage collection, would silently reuse
        a stale symbolic plan (the C++ adapter likewise looks at every block while it flattens)."""
        try:
            import xxhash
            h = xxhash.xxh3_64()
            for a in (r_lambda.col_ptr, r_lambda.row_idx, r_lambda.dim):
                h.update(np.ascontiguousarray(a).view(np.uint8))
            digest = h.intdigest()
        except ImportError:
            import zlib
            digest = 0
            for a in (r_lambda.col_ptr, r_lambda.row_idx, r_lambda.dim):
                digest = zlib.adler32(np.ascontiguousarray(a).view(np.uint8), digest)
        return (r_lambda.nb, r_lambda.nnzb, digest)

    def Solve_PosDef_Blocky(self, r_lambda, r_eta):
        assert r_eta.shape[0] == r_lambda.n, "eta length must equal the matrix dimension"
        if not self._have_symbolic or self._sig is None or self._sig != self._signature(r_lambda):
            self.SymbolicDecomposition_Blocky(r_lambda)
        code, x = self._ctx.factor_solve(r_lambda.vals, r_eta)
        if code == SPP_NOT_POSDEF:
            return False
        r_eta[:] = x
        return True

    def Solve_PosDef(self, r_lambda, r_eta):
        # elementwise entry: symbolic redone on every call (LinearSolver_UberBlock.h:143-258)
        self.Clear_SymbolicDecomposition()
        return self.Solve_PosDef_Blocky(r_lambda, r_eta)

    @property
    def ctx(self):
        return self._ctx
