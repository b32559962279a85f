"""oracle/spp_oracle.py -- TEST INFRASTRUCTURE ONLY (the parity checker).

Python face of the CPU oracle: numpy for the integer/structure work, oracle/libspp_oracle.so
(plain C, oracle/spp_oracle.c) for the floating-point loops, and -- when it has been built in the
development container -- oracle/_ref/libspp_ref.so, i.e. the REFERENCE's own solvers compiled from
/root/reference by oracle/Makefile, driven through oracle/ref_driver.cpp.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (slam_plus_plus_amd/) never does.

Restated reference functions (file:line under /root/reference):
  lambda_structure   CLambdaOps2::AddEntriesInSparseSystem + Alloc_HessianBlocks_v2
                     include/slam/NonlinearSolver_Lambda_Base.h:1852-1931, include/slam/BaseTypes_Binary.h:525-660
  assemble           Calculate_Hessians_v2 + ReduceAll + unary factor (+ LM damping)
                     BaseTypes_Binary.h:759-848, _Lambda_Base.h:563-607,152-197,1903-1924, NonlinearSolver_Lambda_LM.h:228-239
  solve_blocky       CLinearSolver_UberBlock::Solve_PosDef_Blocky, include/slam/LinearSolver_UberBlock.h:312-426
  schur_solve        CLinearSolver_Schur::Solve_PosDef_Blocky, include/slam/LinearSolver_Schur.h:1623-1935
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libspp_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libspp_ref.so")

import sys
sys.path.insert(0, os.path.dirname(_HERE))
from slam_plus_plus_amd.blockcsc import BlockCSC, structure_from_pairs  # noqa: E402  (data model only)

_lib = None
_ref = None
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def build():
    """compile the C restatement (and the reference build when /root/reference is present)"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle"])
    if os.path.exists("/root/reference/src/slam/BlockMatrix.cpp"):
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])
        # the other reference-side test programs (drop-in drivers, the reference's applications with and without the shim,
        # the Lambda recorder): they link libspp_hip.so, so only once that exists; a failure here must not take the build
        # check down (they are checkers: the GPU tests skip what is absent). Incremental: a no-op when up to date, a few
        # minutes from scratch.
        if os.path.exists(os.path.join(_HERE, "..", "slam_plus_plus_amd", "libspp_hip.so")):
            try:
                subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "dropin", "lambda_dump", "apps"], timeout=1500)
            except Exception as e:  # noqa: BLE001
                print("oracle: reference-side test programs not (re)built: %r" % (e,))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle"])
        L = ctypes.CDLL(_LIB)
        L.orc_chol_symbolic.restype = i64
        for f in ("orc_cholesky", "orc_utsolve", "orc_usolve", "orc_dense_llt_solve", "orc_schur_solve"):
            getattr(L, f).restype = ctypes.c_int
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = ctypes.CDLL(_REF)
        R.ref_create.restype = vp
        R.ref_create.argtypes = [ctypes.c_int, ctypes.c_int]
        R.ref_destroy.argtypes = [vp]
        R.ref_set_structure.argtypes = [vp, i64, vp, vp, vp]
        R.ref_solve.argtypes = [vp, vp, vp, vp, vp]
        R.ref_block_ordering.argtypes = [vp, vp]
        R.ref_factor_fill.argtypes = [vp, vp, vp]
        _ref = R
    return _ref


# --------------------------------------------------------------------------------------------------
# assembly
# --------------------------------------------------------------------------------------------------
def lambda_structure(prob):
    """Block structure of Lambda from the graph: every vertex gets its diagonal block, every edge the
    upper off-diagonal block (min id, max id). Returns (BlockCSC, off-diag block index per edge,
    diagonal block index per vertex, reversed flag per edge)."""
    v0 = np.asarray(prob.v0, dtype=np.int64)
    v1 = np.asarray(prob.v1, dtype=np.int64)
    assert np.all(v0 != v1)
    rev = v0 > v1
    st, eblk, dblk = structure_from_pairs(prob.dim, np.minimum(v0, v1), np.maximum(v0, v1))
    return st, eblk, dblk, rev


def assemble(prob, damping=None, weights=None):
    """Lambda (upper block triangle) and eta from per-edge Jacobians, in the reference's order.
    weights: one robust weight per edge (the reference's b_is_robust_edge branch, BaseTypes_Binary.h:768-848)."""
    L = lib()
    st, eblk, dblk, rev = lambda_structure(prob)
    ne, d0, d1, rd = prob.v0.size, prob.d0, prob.d1, prob.rd
    J0 = np.ascontiguousarray(prob.J0, dtype=np.float64)
    J1 = np.ascontiguousarray(prob.J1, dtype=np.float64)
    Om = np.ascontiguousarray(prob.Om, dtype=np.float64)
    r = np.ascontiguousarray(prob.r, dtype=np.float64)
    H01 = np.empty(ne * d0 * d1)
    H00 = np.empty(ne * d0 * d0)
    H11 = np.empty(ne * d1 * d1)
    g0 = np.empty(ne * d0)
    g1 = np.empty(ne * d1)
    rev8 = np.ascontiguousarray(rev, dtype=np.uint8)
    if weights is None:
        L.orc_edge_hessians(d0, d1, rd, i64(ne), _p(J0), _p(J1), _p(Om), _p(r), _p(rev8),
                            _p(H01), _p(H00), _p(H11), _p(g0), _p(g1))
    else:
        wts = np.ascontiguousarray(weights, dtype=np.float64)
        assert wts.shape == (ne,)
        L.orc_edge_hessians_w(d0, d1, rd, i64(ne), _p(J0), _p(J1), _p(Om), _p(r), _p(wts), _p(rev8),
                              _p(H01), _p(H00), _p(H11), _p(g0), _p(g1))
    nv = prob.dim.size
    dim = np.asarray(prob.dim, dtype=np.int64)
    # one source pool: [H01 | H00 | H11 | unary identity]
    src = np.concatenate([H01, H00, H11, np.eye(int(dim[prob.unary_vertex])).ravel() if prob.unary_vertex >= 0 else np.zeros(0)])
    o00, o11, ouf = H01.size, H01.size + H00.size, H01.size + H00.size + H11.size
    e = np.arange(ne, dtype=np.int64)
    # reduction lists in edge order: (destination block, source offset); within an edge the
    # reference registers the off-diagonal block, then vertex 0, then vertex 1
    dst = np.concatenate([eblk, dblk[prob.v0], dblk[prob.v1]])
    soff = np.concatenate([e * d0 * d1, o00 + e * d0 * d0, o11 + e * d1 * d1])
    seq = np.concatenate([3 * e, 3 * e + 1, 3 * e + 2])
    if prob.unary_vertex >= 0:  # added after all edges (_Lambda_Base.h:1903-1924; vertex 0 in the default build, FlatSystem.h:337)
        dst = np.append(dst, dblk[prob.unary_vertex])
        soff = np.append(soff, ouf)
        seq = np.append(seq, 3 * ne)
    order = np.lexsort((seq, dst))
    dst, soff = dst[order], soff[order]
    list_ptr = np.zeros(st.nnzb + 1, dtype=np.int64)
    np.add.at(list_ptr, dst + 1, 1)
    np.cumsum(list_ptr, out=list_ptr)
    blen = (st.dim[st.row_idx].astype(np.int64) * st.dim[st.col_idx]).astype(np.int32)
    vals = np.zeros(st.nvals)
    L.orc_reduce(i64(st.nnzb), _p(st.blk_off), _p(blen), _p(list_ptr), _p(np.ascontiguousarray(soff)), _p(src), _p(vals))
    # eta: per vertex, g0 of edges where it is vertex 0 and g1 where it is vertex 1, in edge order
    gsrc = np.concatenate([g0, g1])
    vdst = np.concatenate([prob.v0, prob.v1])
    gso = np.concatenate([e * d0, g0.size + e * d1])
    gseq = np.concatenate([2 * e, 2 * e + 1])
    order = np.lexsort((gseq, vdst))
    vdst, gso = vdst[order], gso[order]
    vptr = np.zeros(nv + 1, dtype=np.int64)
    np.add.at(vptr, vdst + 1, 1)
    np.cumsum(vptr, out=vptr)
    eta = np.zeros(st.n)
    L.orc_reduce(i64(nv), _p(st.base[:-1].copy()), _p(np.ascontiguousarray(prob.dim, dtype=np.int32)), _p(vptr),
                 _p(np.ascontiguousarray(gso)), _p(gsrc), _p(eta))
    lam = st.with_vals(vals)
    damping = prob.damping if damping is None else damping
    if damping:
        add_damping(lam, damping)
    return lam, eta


def add_damping(lam, alpha):
    """Lambda_ii.diagonal() += alpha (NonlinearSolver_Lambda_LM.h:228-239)"""
    for j in range(lam.nb):
        p = lam.col_ptr[j + 1] - 1
        d = int(lam.dim[j])
        o = int(lam.blk_off[p])
        lam.vals[o:o + d * d:d + 1] += alpha


# --------------------------------------------------------------------------------------------------
# sparse block Cholesky solve (the CLinearSolver_UberBlock path) with a given elimination order
# --------------------------------------------------------------------------------------------------
def permute_upper(lam, order):
    """P Lambda P^T keeping the upper triangle (BlockMatrix.cpp:8183-8349). order[k] = source block
    column that becomes column k. Returns the permuted BlockCSC."""
    L = lib()
    order = np.asarray(order, dtype=np.int64)
    inv = np.empty_like(order)
    inv[order] = np.arange(order.size)
    pr, pc = inv[lam.row_idx], inv[lam.col_idx]
    tr = pr > pc
    nr, nc_ = np.where(tr, pc, pr), np.where(tr, pr, pc)
    pdim = lam.dim[order]
    st, blk, _, = structure_from_pairs(pdim, nr, nc_)[:3]
    rows = lam.dim[lam.row_idx].astype(np.int32)
    cols = lam.dim[lam.col_idx].astype(np.int32)
    vals = np.zeros(st.nvals)
    L.orc_permute_values(i64(lam.nnzb), _p(lam.blk_off), _p(np.ascontiguousarray(st.blk_off[blk])), _p(rows), _p(cols),
                         _p(np.ascontiguousarray(tr, dtype=np.uint8)), _p(lam.vals), _p(vals))
    return st.with_vals(vals)


def etree(lam):
    parent = np.empty(lam.nb, dtype=np.int64)
    lib().orc_etree(i64(lam.nb), _p(lam.col_ptr), _p(lam.row_idx), _p(parent))
    return parent


def cholesky(lam):
    """R with R^T R = lam (upper block triangle). Returns (BlockCSC R, status)."""
    L = lib()
    parent = etree(lam)
    r_col_ptr = np.empty(lam.nb + 1, dtype=np.int64)
    nnzb = L.orc_chol_symbolic(i64(lam.nb), _p(lam.col_ptr), _p(lam.row_idx), _p(parent), _p(r_col_ptr), None)
    r_row_idx = np.empty(nnzb, dtype=np.int64)
    L.orc_chol_symbolic(i64(lam.nb), _p(lam.col_ptr), _p(lam.row_idx), _p(parent), _p(r_col_ptr), _p(r_row_idx))
    cols = np.repeat(np.arange(lam.nb, dtype=np.int64), np.diff(r_col_ptr))
    size = lam.dim[r_row_idx].astype(np.int64) * lam.dim[cols]
    r_off = np.zeros(nnzb, dtype=np.int64)
    np.cumsum(size[:-1], out=r_off[1:])
    r_vals = np.zeros(int(size.sum()))
    st = L.orc_cholesky(i64(lam.nb), _p(lam.dim), _p(lam.col_ptr), _p(lam.row_idx), _p(lam.blk_off), _p(lam.vals),
                        _p(parent), _p(r_col_ptr), _p(r_row_idx), _p(r_off), _p(r_vals))
    return BlockCSC(lam.dim, r_col_ptr, r_row_idx, r_off, r_vals), st


def solve_blocky(lam, eta, order=None):
    """Delta-x = Lambda^-1 eta by permute + up-looking block Cholesky + two triangular solves.
    Returns (status, x): status 0 ok, 1 not positive definite."""
    L = lib()
    if order is None:
        order = np.arange(lam.nb, dtype=np.int64)
    order = np.asarray(order, dtype=np.int64)
    perm = permute_upper(lam, order)
    R, st = cholesky(perm)
    if st:
        return st, None
    # InversePermute_LeftHandSide_Vector: x_perm[block inv[i]] = eta[block i]
    xp = np.concatenate([eta[lam.base[b]:lam.base[b + 1]] for b in order]) if lam.nb < 20000 else _gather(eta, lam, order)
    if L.orc_utsolve(i64(R.nb), _p(R.dim), _p(R.base), _p(R.col_ptr), _p(R.row_idx), _p(R.blk_off), _p(R.vals), _p(xp)):
        return 1, None
    if L.orc_usolve(i64(R.nb), _p(R.dim), _p(R.base), _p(R.col_ptr), _p(R.row_idx), _p(R.blk_off), _p(R.vals), _p(xp)):
        return 1, None
    x = np.empty_like(eta)
    pos = 0
    for b in order:
        d = int(lam.dim[b])
        x[lam.base[b]:lam.base[b] + d] = xp[pos:pos + d]
        pos += d
    return 0, x


def _gather(eta, lam, order):
    idx = np.concatenate([np.arange(lam.base[b], lam.base[b + 1]) for b in order])
    return np.ascontiguousarray(eta[idx])


# --------------------------------------------------------------------------------------------------
# Schur complement solve (the CLinearSolver_Schur path)
# --------------------------------------------------------------------------------------------------
def guided_order(lam):
    """poses (larger width) first, landmarks last, stable (LinearSolver_Schur.cpp:771-838)"""
    dl = int(lam.dim.min())
    poses = np.flatnonzero(lam.dim != dl)
    lms = np.flatnonzero(lam.dim == dl)
    return np.concatenate([poses, lms]).astype(np.int64), poses.size


def schur_solve(lam, eta, want_S=False):
    L = lib()
    order, nc = guided_order(lam)
    perm = permute_upper(lam, order)
    idx = np.concatenate([np.arange(lam.base[b], lam.base[b + 1]) for b in order])
    xp = np.ascontiguousarray(eta[idx])
    S = np.zeros((int(perm.base[nc]),) * 2, order="F") if want_S else None
    st = L.orc_schur_solve(i64(perm.nb), i64(nc), _p(perm.dim), _p(perm.base), _p(perm.col_ptr), _p(perm.row_idx),
                           _p(perm.blk_off), _p(perm.vals), _p(xp), _p(S))
    if st:
        return st, None, S
    x = np.empty_like(eta)
    x[idx] = xp
    return 0, x, S


# --------------------------------------------------------------------------------------------------
# the reference itself (oracle/_ref): 0 UberBlock, 1 CSparse, 2 CHOLMOD, 3 Schur + dense LLT
# --------------------------------------------------------------------------------------------------
class RefSolver:
    BACKENDS = {"uberblock": 0, "csparse": 1, "cholmod": 2, "schur": 3}

    def __init__(self, backend, lam):
        R = ref()
        d = int(lam.dim.max())
        two = np.unique(lam.dim).size == 2
        problem = 2 if two else (0 if d == 3 else 1)
        self.h = R.ref_create(self.BACKENDS[backend], problem)
        self.lam = lam
        st = R.ref_set_structure(self.h, i64(lam.nb), _p(lam.col_ptr), _p(lam.row_idx), _p(lam.dim))
        if st:
            raise RuntimeError("ref_set_structure failed: %d" % st)

    def solve(self, vals, eta):
        x = np.array(eta, dtype=np.float64, copy=True)
        sec = ctypes.c_double()
        st = ref().ref_solve(self.h, _p(np.ascontiguousarray(vals)), _p(self.lam.blk_off), _p(x), ctypes.byref(sec))
        if st < 0:
            raise RuntimeError("ref_solve failed: %d" % st)
        return st, x, sec.value

    def ordering(self):
        out = np.empty(self.lam.nb, dtype=np.int64)
        ref().ref_block_ordering(self.h, _p(out))
        return out

    def close(self):
        if self.h:
            ref().ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------------------------------
# multi-GPU algebra: partial Schur complement of one landmark shard. Landmarks are dealt
# round-robin over the ranks in ascending block-column order (the rule of build_schur_plan() in
# slam_plus_plus_amd/csrc/spp_symbolic.cpp); rank 0 alone carries A and the pose part of eta.
# --------------------------------------------------------------------------------------------------
def landmark_shard(lam, rank, world):
    dl = int(lam.dim.min())
    lms = np.flatnonzero(lam.dim == dl)
    return lms[np.arange(lms.size) % world == rank]


def schur_partial(lam, eta, rank, world):
    """Returns (S_partial (n_p x n_p, upper), x_partial (n_p), pose scalar index, shard landmark blocks)."""
    L = lib()
    dl = int(lam.dim.min())
    poses = np.flatnonzero(lam.dim != dl)
    mine = landmark_shard(lam, rank, world)
    keep = np.concatenate([poses, mine]).astype(np.int64)
    newid = -np.ones(lam.nb, dtype=np.int64)
    newid[keep] = np.arange(keep.size)
    sel = (newid[lam.row_idx] >= 0) & (newid[lam.col_idx] >= 0)
    # sub-matrix in (poses | shard landmarks) order: guided order keeps relative ids, so a block
    # (i, j) with i < j may flip when i is a landmark and j a pose
    r, c = newid[lam.row_idx[sel]], newid[lam.col_idx[sel]]
    tr = r > c
    nr, nc_ = np.where(tr, c, r), np.where(tr, r, c)
    sdim = lam.dim[keep]
    st, blk, _ = structure_from_pairs(sdim, nr, nc_)
    vals = np.zeros(st.nvals)
    rows = lam.dim[lam.row_idx[sel]].astype(np.int32)
    cols = lam.dim[lam.col_idx[sel]].astype(np.int32)
    L.orc_permute_values(i64(int(sel.sum())), _p(np.ascontiguousarray(lam.blk_off[sel])),
                         _p(np.ascontiguousarray(st.blk_off[blk])), _p(rows), _p(cols),
                         _p(np.ascontiguousarray(tr, dtype=np.uint8)), _p(lam.vals), _p(vals))
    sub = st.with_vals(vals)
    idx = np.concatenate([np.arange(lam.base[b], lam.base[b + 1]) for b in keep])
    x = np.ascontiguousarray(eta[idx])
    n_p = int(sub.base[poses.size])
    if rank != 0:  # A and the pose rhs belong to rank 0
        for j in range(poses.size):
            for p in range(sub.col_ptr[j], sub.col_ptr[j + 1]):
                d = int(sub.dim[sub.row_idx[p]]) * int(sub.dim[j])
                sub.vals[sub.blk_off[p]:sub.blk_off[p] + d] = 0
        x[:n_p] = 0
    S = np.zeros((n_p, n_p), order="F")
    L.orc_schur_solve(i64(sub.nb), i64(poses.size), _p(sub.dim), _p(sub.base), _p(sub.col_ptr), _p(sub.row_idx),
                      _p(sub.blk_off), _p(sub.vals), _p(x), _p(S))
    # x[:n_p] now holds the partial reduced rhs if the (partial, possibly indefinite) LLT failed,
    # or the solution of the partial system otherwise -> recompute it explicitly to be safe
    xr = np.ascontiguousarray(eta[idx])
    if rank != 0:
        xr[:n_p] = 0
    xred = xr[:n_p].copy()
    for j in range(poses.size, sub.nb):
        pe = sub.col_ptr[j + 1] - 1
        C = sub.vals[sub.blk_off[pe]:sub.blk_off[pe] + dl * dl].reshape(dl, dl).T
        Ci = np.empty(dl * dl)
        if dl == 3:
            L.orc_inverse3(_p(np.ascontiguousarray(C.T.ravel())), _p(Ci))
            Cinv = Ci.reshape(3, 3).T
        else:
            Cinv = np.linalg.inv(C)
        lj = xr[sub.base[j]:sub.base[j] + dl]
        for p in range(sub.col_ptr[j], pe):
            i = int(sub.row_idx[p])
            di = int(sub.dim[i])
            U = sub.vals[sub.blk_off[p]:sub.blk_off[p] + di * dl].reshape(dl, di).T
            xred[sub.base[i]:sub.base[i] + di] += (U @ (-Cinv)) @ lj
    pose_idx = np.concatenate([np.arange(lam.base[b], lam.base[b + 1]) for b in poses])
    return S, xred, pose_idx, mine


def schur_backsubstitute(lam, eta, dx_poses, blocks):
    """dl = C^-1 (l - U^T dx) for the given landmark blocks (LinearSolver_Schur.h:1867-1881)"""
    dl = int(lam.dim.min())
    poses = np.flatnonzero(lam.dim != dl)
    pbase = np.zeros(lam.nb, dtype=np.int64)
    pbase[poses] = np.concatenate([[0], np.cumsum(lam.dim[poses])[:-1]])
    out = {}
    # blocks touching landmark b: column b (poses with smaller id) and row b in pose columns (larger id)
    by_lm = {int(b): [] for b in blocks}
    for p in range(lam.nnzb):
        i, j = int(lam.row_idx[p]), int(lam.col_idx[p])
        if i == j:
            continue
        if j in by_lm and lam.dim[i] != dl:
            by_lm[j].append((i, p, False))
        elif i in by_lm and lam.dim[j] != dl:
            by_lm[i].append((j, p, True))
    for b in blocks:
        b = int(b)
        pe = lam.col_ptr[b + 1] - 1
        C = lam.vals[lam.blk_off[pe]:lam.blk_off[pe] + dl * dl].reshape(dl, dl).T
        t = eta[lam.base[b]:lam.base[b] + dl].copy()
        for (pose, p, transposed) in by_lm[b]:
            dp = int(lam.dim[pose])
            blk = lam.vals[lam.blk_off[p]:lam.blk_off[p] + dp * dl]
            # U is dp x dl. Stored as is (column-major: element (r, q) at r + dp q), or transposed
            # when the landmark id is smaller (dl x dp column-major: element (q, r) at q + dl r)
            U = blk.reshape(dp, dl) if transposed else blk.reshape(dl, dp).T
            t -= U.T @ dx_poses[pbase[pose]:pbase[pose] + dp]
        out[b] = np.linalg.solve(C, t)
    return out
