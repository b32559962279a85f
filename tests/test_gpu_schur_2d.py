"""The Schur path on 2D pose + 2D landmark SLAM (block widths {3, 2}: the victoria-park / cityTrees10k shape of
SURVEY 8f-4): poses connected by odometry (off-diagonal pose-pose blocks = a sparse A), landmarks observed from
a few poses each. Guided ordering (LinearSolver_Schur.cpp:771-838) splits by width; both reduced-system variants
(dense S / sparse S) are checked against the CPU oracle's sparse block Cholesky of the whole Lambda and, when
oracle/_ref is present, the reference's CSparse and CHOLMOD backends (its fixed-block-size UberBlock instance in
oracle/ref_driver.cpp is compiled for the 3x3 / 6x6 / 6x3 block lists only). Lambda is built on the host from random
well-conditioned Jacobians (two edge groups: (3,3,3) odometry and (3,2,2) observations)."""
import numpy as np
import pytest

from slam_plus_plus_amd import api
from slam_plus_plus_amd.blockcsc import structure_from_pairs
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _problem(n_poses, n_lm, seed, interleave):
    rng = np.random.default_rng(seed)
    nv = n_poses + n_lm
    ids = rng.permutation(nv) if interleave else np.arange(nv)
    pose_id, lm_id = ids[:n_poses], ids[n_poses:]
    dim = np.empty(nv, dtype=np.int32)
    dim[pose_id], dim[lm_id] = 3, 2
    base = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum(dim, out=base[1:])
    n = int(base[-1])
    L = np.zeros((n, n))
    eta = np.zeros(n)
    pairs = []

    def add_edge(a, b, da, db, rd):
        Ja, Jb = rng.normal(size=(rd, da)), rng.normal(size=(rd, db))
        r = rng.normal(size=rd)
        sa, sb = slice(base[a], base[a] + da), slice(base[b], base[b] + db)
        L[sa, sa] += Ja.T @ Ja
        L[sb, sb] += Jb.T @ Jb
        L[sa, sb] += Ja.T @ Jb
        L[sb, sa] += Jb.T @ Ja
        eta[sa] += Ja.T @ r
        eta[sb] += Jb.T @ r
        pairs.append((min(a, b), max(a, b)))

    for i in range(n_poses - 1):                      # odometry chain + a few loop closures
        add_edge(pose_id[i], pose_id[i + 1], 3, 3, 3)
    for _ in range(n_poses // 5):
        i, j = sorted(rng.choice(n_poses, size=2, replace=False))
        add_edge(pose_id[i], pose_id[j], 3, 3, 3)
    for l in range(n_lm):                             # every landmark seen from 2..5 poses around a centre pose
        c = rng.integers(0, n_poses)
        for p in np.unique(np.clip(c + rng.integers(-6, 7, size=rng.integers(2, 6)), 0, n_poses - 1)):
            add_edge(pose_id[p], lm_id[l], 3, 2, 2)
    L += 1e-2 * np.eye(n)                             # the unary factor's role: no gauge freedom
    rows, cols = np.array(pairs).T
    st, _, _ = structure_from_pairs(dim, rows, cols)
    vals = np.zeros(st.nvals)
    for j in range(st.nb):
        for p in range(st.col_ptr[j], st.col_ptr[j + 1]):
            i = st.row_idx[p]
            blk = L[base[i]:base[i] + dim[i], base[j]:base[j] + dim[j]]
            vals[st.blk_off[p]:st.blk_off[p] + blk.size] = blk.ravel(order="F")
    return st.with_vals(vals), eta


@pytest.mark.parametrize("n_poses,n_lm,interleave", [(60, 90, False), (200, 500, False), (150, 300, True)])
def test_pose_landmark_2d_schur_dense_and_sparse_reduced_system(n_poses, n_lm, interleave):
    lam, eta = _problem(n_poses, n_lm, 7 + n_poses, interleave)
    st, xo = orc.solve_blocky(lam, eta)
    assert st == 0
    xs = {}
    for mode in (api.MODE_AUTO, api.MODE_SCHUR, api.MODE_SCHUR_SPARSE, api.MODE_SPARSE):
        solver = api.CLinearSolver_HIP(mode=mode)
        x = eta.copy()
        assert solver.Solve_PosDef_Blocky(lam, x)
        if mode == api.MODE_AUTO:
            assert solver.ctx.info("MODE") == api.MODE_SCHUR            # two widths, landmarks not connected: guided Schur
            assert solver.ctx.info("N_REDUCED") == 3 * n_poses and solver.ctx.info("N_LANDMARKS") == n_lm
        assert _rel(x, xo) < TOL, (mode, _rel(x, xo))
        assert np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta) < 1e-12
        xs[mode] = x
    assert _rel(xs[api.MODE_SCHUR_SPARSE], xs[api.MODE_SCHUR]) < 1e-12
    if orc.have_ref():
        for be in ("csparse", "cholmod"):
            rs = orc.RefSolver(be, lam)
            code, xr, _ = rs.solve(lam.vals, eta)
            assert code == 0 and _rel(xs[api.MODE_SCHUR], xr) < TOL, be
            rs.close()


@pytest.mark.parametrize("name", ["lm2d_small", "lm2d_interleaved"])
def test_range_bearing_group_assembled_and_solved_on_the_device(name):
    """the (3, 2, 2) edge group end to end: device assembly (spp_assemble_device), guided Schur with 3-wide poses
    and 2-wide landmarks, against the oracle's assembly + sparse block Cholesky"""
    from slam_plus_plus_amd import synth
    prob = synth.make(name)
    lam_o, eta_o = orc.assemble(prob)
    code, xo = orc.solve_blocky(lam_o, eta_o)
    assert code == 0
    ctx = api.Context(0)
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, 3, 2, 2, prob.unary_vertex)
    arrs = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    dv, de = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n)
    ctx.assemble_device(*[a.ptr for a in arrs], prob.damping, dv.ptr, de.ptr)
    assert np.abs(dv.download() - lam_o.vals).max() <= 1e-13 * np.abs(lam_o.vals).max()
    for mode in (api.MODE_AUTO, api.MODE_SCHUR_SPARSE, api.MODE_SPARSE):
        ctx.analyze(st, mode)
        if mode == api.MODE_AUTO:
            assert ctx.info("MODE") == api.MODE_SCHUR
        dr = api.DeviceArray.from_host(ctx, de.download())
        assert ctx.factor_solve_device(dv.ptr, dr.ptr) == 0
        assert _rel(dr.download(), xo) < TOL
    ctx.close()
