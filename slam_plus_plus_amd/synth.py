"""Seeded synthetic problems shaped like the five BASELINE.json configs. numpy only.

The real datasets (manhattanOlson3500, sphere2500, BAL Ladybug-49, Venice-871) are not available
offline, so every config is a *shape-matched synthetic*: same vertex/edge counts, same block
widths, plausible geometry. A problem is the INPUT of the hot path: one homogeneous group of binary
edges with per-edge Jacobians J0 (rd x d0), J1 (rd x d1), information Omega (rd x rd) and residual r
-- i.e. the outputs of the reference's Calculate_Jacobians_Expectation_Error
(include/slam/BaseTypes_Binary.h:763-765), which stays reference code.
All per-edge arrays are column-major blocks flattened edge by edge.
"""
import numpy as np


class Problem(dict):
    """dict with attribute access: dim, v0, v1, d0, d1, rd, J0, J1, Om, r, unary_vertex, damping, name.
    unary_vertex: the vertex whose diagonal block receives the unit unary factor -- vertex 0 in the reference's
    default build (__AUTO_UNARY_FACTOR_ON_VERTEX_ZERO, include/slam/FlatSystem.h:331-337; pinned by
    tests/golden/ba_lambda.npz, where vertex 0 is a landmark), whatever its type."""
    __getattr__ = dict.__getitem__


# ------------------------------------------------------------------------------------------------
# bundle adjustment (configs 3, 4, 5): cameras (6) + points (3), 2-d reprojection residuals
# ------------------------------------------------------------------------------------------------
def _track_lengths(rng, npts, nobs, kmin, kmax, heavy_tail):
    mean = nobs / float(npts)
    if heavy_tail:
        k = kmin + rng.geometric(1.0 / (mean - kmin + 1.0), size=npts) - 1
    else:
        lo = int(np.floor(mean))
        k = np.full(npts, lo, dtype=np.int64)
        k[rng.permutation(npts)[:nobs - lo * npts]] += 1
    k = np.clip(k, kmin, kmax).astype(np.int64)
    diff = int(nobs - k.sum())
    while diff != 0:  # nudge random tracks until the observation count is exact
        step = 1 if diff > 0 else -1
        ok = np.flatnonzero((k + step >= kmin) & (k + step <= kmax))
        pick = rng.choice(ok, size=min(abs(diff), ok.size), replace=False)
        k[pick] += step
        diff = int(nobs - k.sum())
    return k


def ba_problem(nc, npts, nobs, seed, heavy_tail=True, interleave=False, spread=0.12, name="ba"):
    """nc cameras on a circle looking at the origin, npts points in a cube, exactly nobs observations.
    Point j is seen by k_j distinct cameras drawn around a centre camera (window ~ spread * nc), which
    yields a banded-to-dense reduced camera system. interleave=True shuffles vertex ids so that about
    half of the camera-point blocks have (point id < camera id) and are stored transposed
    (reference BaseTypes_Binary.h:783-806)."""
    rng = np.random.default_rng(seed)
    kmax = min(nc, 64)
    k = _track_lengths(rng, npts, nobs, 2, kmax, heavy_tail)
    centre = rng.integers(0, nc, size=npts)
    cam_of = np.empty(nobs, dtype=np.int64)
    pt_of = np.repeat(np.arange(npts, dtype=np.int64), k)
    start = np.zeros(npts + 1, dtype=np.int64)
    np.cumsum(k, out=start[1:])
    half = max(2, int(spread * nc))
    for kk in np.unique(k):
        idx = np.flatnonzero(k == kk)
        win = min(nc, max(2 * half + 1, int(kk)))
        # kk distinct offsets in a window: argsort of random keys
        offs = np.argsort(rng.random((idx.size, win)), axis=1)[:, :kk] - win // 2
        cams = np.sort((centre[idx, None] + offs) % nc, axis=1)
        pos = start[idx, None] + np.arange(kk)[None, :]
        cam_of[pos.ravel()] = cams.ravel()
    # geometry
    th = 2 * np.pi * np.arange(nc) / nc
    C = np.stack([10 * np.cos(th), 10 * np.sin(th), 0.5 * np.sin(3 * th)], axis=1)
    z = -C / np.linalg.norm(C, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up[None, :], z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=1)  # rows = camera axes (world -> camera)
    X = rng.uniform(-2, 2, size=(npts, 3))
    pc = np.einsum("eij,ej->ei", R[cam_of], X[pt_of] - C[cam_of])
    f = 500.0
    iz = 1.0 / pc[:, 2]
    Kp = np.zeros((nobs, 2, 3))
    Kp[:, 0, 0] = f * iz
    Kp[:, 1, 1] = f * iz
    Kp[:, 0, 2] = -f * pc[:, 0] * iz * iz
    Kp[:, 1, 2] = -f * pc[:, 1] * iz * iz
    Jp = np.einsum("eij,ejk->eik", Kp, R[cam_of])            # d(u,v)/dX
    skew = np.zeros((nobs, 3, 3))
    skew[:, 0, 1], skew[:, 0, 2] = -pc[:, 2], pc[:, 1]
    skew[:, 1, 0], skew[:, 1, 2] = pc[:, 2], -pc[:, 0]
    skew[:, 2, 0], skew[:, 2, 1] = -pc[:, 1], pc[:, 0]
    Jc = np.concatenate([Kp, -np.einsum("eij,ejk->eik", Kp, skew)], axis=2)  # d(u,v)/d(t, w)
    r = rng.normal(0, 0.5, size=(nobs, 2)) + 0.02 * f * iz[:, None] * rng.normal(0, 1, size=(nobs, 2))
    Om = np.tile(np.eye(2).ravel(), (nobs, 1))
    # vertex ids
    nv = nc + npts
    if interleave:
        perm = rng.permutation(nv)
        cam_id, pt_id = perm[:nc], perm[nc:]
    else:
        cam_id, pt_id = np.arange(nc), nc + np.arange(npts)
    dim = np.empty(nv, dtype=np.int32)
    dim[cam_id] = 6
    dim[pt_id] = 3
    # Levenberg-Marquardt damping as the reference applies it on BA inputs (slam_app switches to LM,
    # src/slam_app/Main.cpp:203-208): alpha0 = 1e-3 * largest diagonal entry of any vertex Hessian
    # (NonlinearSolver_Lambda_LM.h:151-199)
    h0 = np.einsum("eri,eri->ei", Jc, Jc).max()
    h1 = np.einsum("eri,eri->ei", Jp, Jp).max()
    return Problem(name=name, dim=dim, v0=cam_id[cam_of], v1=pt_id[pt_of], d0=6, d1=3, rd=2,
                   J0=np.ascontiguousarray(Jc.transpose(0, 2, 1)).reshape(nobs, 12),  # col-major 2x6
                   J1=np.ascontiguousarray(Jp.transpose(0, 2, 1)).reshape(nobs, 6),   # col-major 2x3
                   Om=Om, r=r, unary_vertex=0, damping=1e-3 * float(max(h0, h1)),
                   nc=nc, npts=npts, geometry=dict(R=R, C=C, X=X, f=f, cam_of=cam_of, pt_of=pt_of, cam_id=cam_id, pt_id=pt_id))


def ba_states(prob):
    """The same scene in the REFERENCE's parameterization, as input of spp_ba_linearize_device: cameras
    [t | axis-angle] (world -> camera, CVertexCam), intrinsics fx fy cx cy k, points, and measurements
    z = projection + the problem's residual (so that the device residual reproduces prob.r).
    Returns dict(cams (nc,6), intr (nc,5), points (np,3), meas (no,2), cam_of, pt_of int32,
    cam_dxoff, pt_dxoff int64: scalar offset of every vertex in the solution vector)."""
    from scipy.spatial.transform import Rotation
    g = prob.geometry
    R, C, X, f = g["R"], g["C"], g["X"], g["f"]
    nc, npts = R.shape[0], X.shape[0]
    cams = np.concatenate([-np.einsum("cij,cj->ci", R, C), Rotation.from_matrix(R).as_rotvec()], axis=1)
    intr = np.tile(np.array([f, f, 0.0, 0.0, 0.0]), (nc, 1))
    pc = np.einsum("eij,ej->ei", R[g["cam_of"]], X[g["pt_of"]] - C[g["cam_of"]])
    uv = f * pc[:, :2] / pc[:, 2:3]
    base = np.zeros(prob.dim.size + 1, dtype=np.int64)
    np.cumsum(prob.dim, out=base[1:])
    return dict(cams=cams, intr=intr, points=X.copy(), meas=uv + prob.r, cam_of=g["cam_of"].astype(np.int32),
                pt_of=g["pt_of"].astype(np.int32), cam_dxoff=base[g["cam_id"]].copy(), pt_dxoff=base[g["pt_id"]].copy())


# ------------------------------------------------------------------------------------------------
# 2D pose graph (config 1): manhattan-world random walk, 3x3 blocks
# ------------------------------------------------------------------------------------------------
def se2_problem(n=3500, n_loops=2099, seed=1234, name="manhattan3500"):
    rng = np.random.default_rng(seed)
    heading = np.zeros(n, dtype=np.int64)
    turn = rng.random(n) < 0.25
    heading[1:] = np.cumsum(np.where(turn[1:], rng.choice([-1, 1], size=n - 1), 0)) % 4
    step = np.stack([np.cos(heading * np.pi / 2), np.sin(heading * np.pi / 2)], axis=1).round()
    t = np.zeros((n, 2))
    t[1:] = np.cumsum(step[:-1], axis=0)
    theta = heading * np.pi / 2
    i0 = np.arange(n - 1)
    i1 = i0 + 1
    # loop closures between non-consecutive poses closer than 1.5
    from scipy.spatial import cKDTree
    pairs = cKDTree(t).query_pairs(1.5, output_type="ndarray")
    pairs = pairs[np.abs(pairs[:, 0] - pairs[:, 1]) > 1]
    pick = rng.permutation(pairs.shape[0])[:n_loops]
    lc = pairs[np.sort(pick)]
    flip = rng.random(lc.shape[0]) < 0.5  # some closures point backwards: exercises the reversed-id path
    a = np.where(flip, lc[:, 1], lc[:, 0])
    b = np.where(flip, lc[:, 0], lc[:, 1])
    v0 = np.concatenate([i0, a])
    v1 = np.concatenate([i1, b])
    ne = v0.size
    # estimate = ground truth + noise (linearization point)
    te = t + rng.normal(0, 0.05, size=t.shape)
    the = theta + rng.normal(0, 0.02, size=n)
    c, s = np.cos(the[v0]), np.sin(the[v0])
    d = te[v1] - te[v0]
    J0 = np.zeros((ne, 3, 3))
    J1 = np.zeros((ne, 3, 3))
    J0[:, 0, 0], J0[:, 0, 1], J0[:, 0, 2] = -c, -s, -s * d[:, 0] + c * d[:, 1]
    J0[:, 1, 0], J0[:, 1, 1], J0[:, 1, 2] = s, -c, -c * d[:, 0] - s * d[:, 1]
    J0[:, 2, 2] = -1
    J1[:, 0, 0], J1[:, 0, 1] = c, s
    J1[:, 1, 0], J1[:, 1, 1] = -s, c
    J1[:, 2, 2] = 1
    # residual = measurement (truth + sensor noise) - prediction at the estimate
    ct, st = np.cos(theta[v0]), np.sin(theta[v0])
    dt = t[v1] - t[v0]
    zmeas = np.stack([ct * dt[:, 0] + st * dt[:, 1], -st * dt[:, 0] + ct * dt[:, 1], theta[v1] - theta[v0]], axis=1)
    zmeas += rng.normal(0, 1, size=zmeas.shape) * np.array([0.03, 0.03, 0.01])
    pred = np.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1], the[v1] - the[v0]], axis=1)
    r = zmeas - pred
    r[:, 2] = (r[:, 2] + np.pi) % (2 * np.pi) - np.pi
    Om = np.tile(np.diag([1111.11, 1111.11, 10000.0]).ravel(), (ne, 1))
    return Problem(name=name, dim=np.full(n, 3, dtype=np.int32), v0=v0, v1=v1, d0=3, d1=3, rd=3,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(ne, 9),
                   J1=np.ascontiguousarray(J1.transpose(0, 2, 1)).reshape(ne, 9),
                   Om=Om, r=r, unary_vertex=0, damping=0.0,
                   geometry=dict(kind="se2", poses=np.concatenate([te, the[:, None]], axis=1), meas=zmeas))


# ------------------------------------------------------------------------------------------------
# 3D pose graph (config 2): sphere, 6x6 blocks
# ------------------------------------------------------------------------------------------------
def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 1, 0], R[..., 1, 1], R[..., 2, 2] = c, -s, s, c, 1
    return R


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    R[..., 0, 0], R[..., 0, 2], R[..., 2, 0], R[..., 2, 2], R[..., 1, 1] = c, s, -s, c, 1
    return R


def _skew(v):
    S = np.zeros(v.shape[:-1] + (3, 3))
    S[..., 0, 1], S[..., 0, 2] = -v[..., 2], v[..., 1]
    S[..., 1, 0], S[..., 1, 2] = v[..., 2], -v[..., 0]
    S[..., 2, 0], S[..., 2, 1] = -v[..., 1], v[..., 0]
    return S


def se3_problem(rings=50, per_ring=50, seed=2500, name="sphere2500"):
    rng = np.random.default_rng(seed)
    n = rings * per_ring
    ring = np.repeat(np.arange(rings), per_ring)
    k = np.tile(np.arange(per_ring), rings)
    az = 2 * np.pi * k / per_ring
    el = np.pi * (ring + 1) / (rings + 1) - np.pi / 2
    t = 50 * np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    R = _rot_z(az) @ _rot_y(-el)
    i0 = np.arange(n - 1)
    i1 = i0 + 1                               # 2499 odometry edges along the spiral
    r0 = np.arange(n - per_ring)
    r1 = r0 + per_ring                        # 2450 ring-to-ring edges
    v0 = np.concatenate([i0, r0])
    v1 = np.concatenate([i1, r1])
    ne = v0.size
    te = t + rng.normal(0, 0.3, size=t.shape)
    Re = R @ (np.eye(3) + _skew(rng.normal(0, 0.02, size=(n, 3))))
    Rt = Re[v0].transpose(0, 2, 1)
    d = np.einsum("eij,ej->ei", Rt, te[v1] - te[v0])
    Rij = Rt @ Re[v1]
    J0 = np.zeros((ne, 6, 6))
    J1 = np.zeros((ne, 6, 6))
    J0[:, :3, :3] = -Rt
    J0[:, :3, 3:] = _skew(d)
    J0[:, 3:, 3:] = -Rij.transpose(0, 2, 1)
    J1[:, :3, :3] = Rt
    J1[:, 3:, 3:] = np.eye(3)
    r = rng.normal(0, 1, size=(ne, 6)) * np.array([0.05, 0.05, 0.05, 0.01, 0.01, 0.01]) + \
        0.1 * rng.normal(0, 1, size=(ne, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    Om = np.tile(np.diag([400.0] * 3 + [1e4] * 3).ravel(), (ne, 1))
    return Problem(name=name, dim=np.full(n, 6, dtype=np.int32), v0=v0, v1=v1, d0=6, d1=6, rd=6,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(ne, 36),
                   J1=np.ascontiguousarray(J1.transpose(0, 2, 1)).reshape(ne, 36),
                   Om=Om, r=r, unary_vertex=0, damping=0.0, geometry=dict(kind="se3", t=t, R=R, te=te, Re=Re, seed=seed))


def landmark2d_problem(n_poses=80, n_lm=200, seed=32, interleave=False, name="lm2d_small"):
    """2D poses (3) observing 2D point landmarks (2): ONE edge group (3, 2, 2) -- the pose-landmark edges of
    victoria-park-style SLAM (CEdgePoseLandmark2D, include/slam/SE2_Types.h; range-bearing Jacobians of
    C2DJacobians::Observation2D_RangeBearing, 2DSolverBase.h:420+). Every pose sees some landmarks and every
    landmark is seen at least twice, so Lambda is positive definite with the unary factor on the first pose."""
    rng = np.random.default_rng(seed)
    nv = n_poses + n_lm
    ids = rng.permutation(nv) if interleave else np.arange(nv)
    pose_id, lm_id = ids[:n_poses], ids[n_poses:]
    P = np.concatenate([rng.uniform(0, 30, size=(n_poses, 2)), rng.uniform(-np.pi, np.pi, size=(n_poses, 1))], axis=1)
    Lm = rng.uniform(0, 30, size=(n_lm, 2))
    from scipy.spatial import cKDTree
    _, near = cKDTree(P[:, :2]).query(Lm, k=4)                    # each landmark: its 4 nearest poses
    po = near.ravel()
    lo = np.repeat(np.arange(n_lm), 4)
    seen = np.zeros(n_poses, dtype=bool)
    seen[po] = True
    extra = np.flatnonzero(~seen)                                 # poses that saw nothing: give them their nearest landmark
    if extra.size:
        _, nl = cKDTree(Lm).query(P[extra, :2], k=2)
        po = np.concatenate([po, np.repeat(extra, 2)])
        lo = np.concatenate([lo, nl.ravel()])
    ne = po.size
    d = Lm[lo] - P[po, :2]
    q = (d ** 2).sum(axis=1)
    sq = np.sqrt(q)
    J0 = np.zeros((ne, 2, 3))                                     # d(range, bearing) / d(x, y, theta)
    J1 = np.zeros((ne, 2, 2))                                     # d(range, bearing) / d(lx, ly)
    J0[:, 0, 0], J0[:, 0, 1] = -d[:, 0] / sq, -d[:, 1] / sq
    J0[:, 1, 0], J0[:, 1, 1], J0[:, 1, 2] = d[:, 1] / q, -d[:, 0] / q, -1
    J1[:, 0, 0], J1[:, 0, 1] = d[:, 0] / sq, d[:, 1] / sq
    J1[:, 1, 0], J1[:, 1, 1] = -d[:, 1] / q, d[:, 0] / q
    dim = np.empty(nv, dtype=np.int32)
    dim[pose_id], dim[lm_id] = 3, 2
    Om = np.tile(np.diag([100.0, 2500.0]).ravel(), (ne, 1))
    r = rng.normal(0, 1, size=(ne, 2)) * np.array([0.1, 0.02])
    return Problem(name=name, dim=dim, v0=pose_id[po], v1=lm_id[lo], d0=3, d1=2, rd=2,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(ne, 6),
                   J1=np.ascontiguousarray(J1.transpose(0, 2, 1)).reshape(ne, 4),
                   Om=Om, r=r, unary_vertex=0, damping=1e-2)


def pose_graph_states(prob):
    """The same pose graph as states + measurements in the REFERENCE's parameterization, as input of
    spp_se2_/se3_linearize_device: poses (n, 3) x y theta or (n, 6) [t | axis-angle] at the noisy estimate,
    meas (ne, 3 or 6) = relative pose of the ground truth + sensor noise, v0 / v1 int32."""
    g = prob.geometry
    if g["kind"] == "se2":
        return dict(dof=3, poses=g["poses"], meas=g["meas"], v0=prob.v0.astype(np.int32), v1=prob.v1.astype(np.int32))
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(g["seed"] + 1)
    Rt, Re = Rotation.from_matrix(g["R"]), Rotation.from_matrix(g["Re"])   # from_matrix projects Re onto SO(3)
    v0, v1 = prob.v0, prob.v1
    zt = Rt[v0].inv().apply(g["t"][v1] - g["t"][v0]) + rng.normal(0, 0.05, size=(v0.size, 3))
    zr = (Rt[v0].inv() * Rt[v1] * Rotation.from_rotvec(rng.normal(0, 0.01, size=(v0.size, 3)))).as_rotvec()
    return dict(dof=6, poses=np.concatenate([g["te"], Re.as_rotvec()], axis=1), meas=np.concatenate([zt, zr], axis=1),
                v0=v0.astype(np.int32), v1=v1.astype(np.int32))


# ------------------------------------------------------------------------------------------------
# the five BASELINE.json configs + small variants for fast tests
# ------------------------------------------------------------------------------------------------
CONFIGS = {
    "manhattan3500": lambda: se2_problem(3500, 2099, 1234),
    "sphere2500": lambda: se3_problem(50, 50, 2500),
    "ladybug49": lambda: ba_problem(49, 7776, 31843, 49, heavy_tail=False, spread=0.06, name="ladybug49"),
    "venice871": lambda: ba_problem(871, 530304, 2838740, 871, heavy_tail=True, name="venice871"),
    "synthetic10k": lambda: ba_problem(10000, 2000000, 10000000, 10000, heavy_tail=False, spread=0.0005, name="synthetic10k"),
    # small cases (seconds on the CPU oracle)
    "ba_tiny": lambda: ba_problem(7, 40, 150, 7, heavy_tail=True, spread=0.5, name="ba_tiny"),
    "ba_small": lambda: ba_problem(30, 1500, 7000, 30, heavy_tail=True, spread=0.2, name="ba_small"),
    "ba_interleaved": lambda: ba_problem(25, 900, 4000, 25, heavy_tail=True, interleave=True, spread=0.3, name="ba_interleaved"),
    "ba_medium": lambda: ba_problem(150, 20000, 100000, 150, heavy_tail=True, name="ba_medium"),
    # a long camera trajectory: every point is seen from a narrow window of cameras, the reduced camera
    # system is banded (the shape of BASELINE config 5 at a size the CPU reference finishes in seconds)
    "ba_banded": lambda: ba_problem(600, 30000, 150000, 600, heavy_tail=False, spread=0.01, name="ba_banded"),
    "lm2d_small": lambda: landmark2d_problem(80, 200, 32),
    "lm2d_interleaved": lambda: landmark2d_problem(60, 150, 33, interleave=True, name="lm2d_interleaved"),
    "se2_small": lambda: se2_problem(300, 150, 12, name="se2_small"),
    "se3_small": lambda: se3_problem(8, 12, 13, name="se3_small"),
}


def make(name):
    return CONFIGS[name]()
