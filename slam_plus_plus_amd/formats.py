"""File formats on either side of the hot path, so that fixtures are interchangeable with
the reference and real datasets can be ingested the moment they are supplied.

* MatrixMarket + block layout pair written by `slam_plus_plus -dsm` (system.mtx / system.bla):
  reference writer src/slam/BlockMatrix.cpp:12063-12205 (Save_BlockLayout, Save_MatrixMarket), reader
  :11589,11682-12060. `.bla` = "rows x cols (nnz)" / "brows x bcols (nblocks)" / row bases + total /
  column bases + total. A symmetric dump lists the stored UPPER triangle as lower-triangle coordinates
  (col+1, row+1, value), values printed with %.15g / %.15f.
* SLAM++ / g2o-style text graphs: the token set of include/slam_app/ParsePrimitives.h:75-1665 that
  the five BASELINE.json configs use (2D poses, 3D poses, BA cameras / points / projections).
  Geometry (Jacobian evaluation) stays with the reference; for 2D pose graphs `se2_linearize`
  provides the analytic Jacobians so that a graph file can be turned into hot-path inputs.
"""
import numpy as np

from .blockcsc import BlockCSC, structure_from_pairs


# --------------------------------------------------------------------------------------------------
# system.mtx / system.bla
# --------------------------------------------------------------------------------------------------
def save_matrix_market(path_mtx, path_bla, lam, kind="lambda"):
    """Write Lambda (upper block triangle) the way CUberBlockMatrix::Save_MatrixMarket(..., 'U') does."""
    di = lam.dim[lam.row_idx].astype(np.int64)
    dj = lam.dim[lam.col_idx].astype(np.int64)
    rows, cols, vals = [], [], []
    for p in range(lam.nnzb):
        i, j = int(lam.row_idx[p]), int(lam.col_idx[p])
        blk = lam.vals[lam.blk_off[p]:lam.blk_off[p] + di[p] * dj[p]].reshape(dj[p], di[p]).T
        r = lam.base[i] + np.arange(di[p])[:, None]
        c = lam.base[j] + np.arange(dj[p])[None, :]
        keep = (c >= r)
        rows.append(np.broadcast_to(r, blk.shape)[keep])
        cols.append(np.broadcast_to(c, blk.shape)[keep])
        vals.append(blk[keep])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    with open(path_mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%-------------------------------------------------------------------------------\n")
        f.write("% UberBlockMatrix matrix dump\n% kind: " + kind + "\n")
        f.write("%-------------------------------------------------------------------------------\n")
        f.write("%d %d %d\n" % (lam.n, lam.n, vals.size))
        for r, c, v in zip(rows, cols, vals):  # lower-triangle coordinates: (col + 1, row + 1)
            f.write(("%d %d %.15f\n" if abs(v) > 1 else "%d %d %.15g\n") % (c + 1, r + 1, v))
    with open(path_bla, "w") as f:
        f.write("%d x %d (%d)\n" % (lam.n, lam.n, int((di * dj).sum())))
        f.write("%d x %d (%d)\n" % (lam.nb, lam.nb, lam.nnzb))
        f.write(" ".join(str(int(b)) for b in lam.base) + "\n")
        f.write(" ".join(str(int(b)) for b in lam.base) + "\n")


def load_block_layout(path_bla):
    with open(path_bla) as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    n = int(lines[0].split()[0])
    nb = int(lines[1].split()[0])
    base = np.array(lines[2].split(), dtype=np.int64)
    assert base.size == nb + 1 and base[-1] == n and base[0] == 0
    return n, nb, base


def load_matrix_market(path_mtx, path_bla):
    """Read a (symmetric or general) MatrixMarket file plus its block layout into an upper BlockCSC.
    Entries below the diagonal of a general file are ignored; every block that holds at least one
    entry is materialized fully (missing entries are zero), diagonal blocks are mirrored."""
    n, nb, base = load_block_layout(path_bla)
    dim = np.diff(base).astype(np.int32)
    rr, cc, vv = [], [], []
    symmetric = False
    with open(path_mtx) as f:
        header = None
        for ln in f:
            if ln.startswith("%%MatrixMarket"):
                symmetric = "symmetric" in ln
                continue
            if ln.startswith("%") or not ln.strip():
                continue
            if header is None:
                header = ln.split()
                assert int(header[0]) == n and int(header[1]) == n
                continue
            a, b, v = ln.split()
            rr.append(int(a) - 1)
            cc.append(int(b) - 1)
            vv.append(float(v))
    rr, cc, vv = np.array(rr, dtype=np.int64), np.array(cc, dtype=np.int64), np.array(vv)
    if symmetric:  # stored as lower-triangle coordinates: flip into the upper triangle
        lo = rr > cc
        rr, cc = np.where(lo, cc, rr), np.where(lo, rr, cc)
    keep = rr <= cc
    rr, cc, vv = rr[keep], cc[keep], vv[keep]
    brow = np.searchsorted(base, rr, side="right") - 1
    bcol = np.searchsorted(base, cc, side="right") - 1
    st, blk, _ = structure_from_pairs(dim, brow, bcol)
    vals = np.zeros(st.nvals)
    di = dim[brow].astype(np.int64)
    pos = st.blk_off[blk] + (rr - base[brow]) + (cc - base[bcol]) * di
    vals[pos] = vv
    # mirror the diagonal blocks (the reference stores them fully symmetric)
    dsel = brow == bcol
    pos_t = st.blk_off[blk[dsel]] + (cc[dsel] - base[bcol[dsel]]) + (rr[dsel] - base[brow[dsel]]) * di[dsel]
    vals[pos_t] = vv[dsel]
    return st.with_vals(vals)


# --------------------------------------------------------------------------------------------------
# text graphs
# --------------------------------------------------------------------------------------------------
_SE2_EDGE = {"EDGE_SE2", "EDGE2", "EDGE", "ODOMETRY"}
_SE2_VERTEX = {"VERTEX_SE2", "VERTEX2", "VERTEX"}
_SE3_EDGE = {"EDGE3", "EDGE_SE3", "EDGE3:AXISANGLE", "EDGE_SE3:AXISANGLE"}
_SE3_VERTEX = {"VERTEX3", "VERTEX_SE3"}


def _upper_to_full(u, d):
    m = np.zeros((d, d))
    m[np.triu_indices(d)] = u
    return m + np.triu(m, 1).T


def load_graph(path):
    """Parse the tokens the BASELINE configs use. Returns a dict of numpy arrays:
      se2_vertices (id, x, y, theta), se2_edges (i, j, dx, dy, dtheta) + se2_info (3x3 each)
      se3_vertices (id, 6: t + roll pitch yaw as in the file), se3_edges (i, j, t, AXIS-ANGLE: the parser's conversion
      of roll-pitch-yaw is applied to EDGE3 / EDGE_SE3, EDGE3:AXISANGLE is taken as it stands) + se3_info (6x6 each)
      cams (id, 6 pose + 5 intrinsics), points (id, xyz), projections (point id, cam id, u, v) + proj_info (2x2)
    2D information is given as the 6 upper-triangular values in the order of
    ParsePrimitives.h (xx xy yy tt xt yt for the classic EDGE2 format is NOT assumed: the g2o
    EDGE_SE2 order xx xy xt yy yt tt is)."""
    out = {k: [] for k in ("se2_vertices", "se2_edges", "se2_info", "se3_vertices", "se3_edges", "se3_info",
                           "cams", "points", "projections", "proj_info")}
    with open(path) as f:
        for ln in f:
            t = ln.split()
            if not t or t[0].startswith("#") or t[0].startswith("%"):
                continue
            tok, a = t[0].upper(), t[1:]
            if tok in _SE2_VERTEX and len(a) >= 4:
                out["se2_vertices"].append([float(x) for x in a[:4]])
            elif tok in _SE2_EDGE and len(a) >= 11:
                out["se2_edges"].append([float(x) for x in a[:5]])
                out["se2_info"].append(_upper_to_full([float(x) for x in a[5:11]], 3))
            elif tok in _SE3_VERTEX and len(a) >= 7:
                out["se3_vertices"].append([float(x) for x in a[:7]])
            elif tok in _SE3_EDGE and len(a) >= 29:
                m = [float(x) for x in a[:8]]
                if not tok.endswith(":AXISANGLE"):
                    # roll-pitch-yaw -> axis-angle exactly as the parser does (ParsePrimitives.h:504-519): Q = Rz Ry Rx
                    from scipy.spatial.transform import Rotation
                    m[5:8] = Rotation.from_euler("ZYX", [m[7], m[6], m[5]]).as_rotvec().tolist()
                out["se3_edges"].append(m)
                out["se3_info"].append(_upper_to_full([float(x) for x in a[8:29]], 6))
            elif tok == "VERTEX_CAM" and len(a) >= 13:
                out["cams"].append([float(x) for x in a[:13]])
            elif tok == "VERTEX_XYZ" and len(a) >= 4:
                out["points"].append([float(x) for x in a[:4]])
            elif tok in ("EDGE_PROJECT_P2MC", "EDGE_PROJECT_P2C") and len(a) >= 7:
                out["projections"].append([float(x) for x in a[:4]])
                out["proj_info"].append(_upper_to_full([float(x) for x in a[4:7]], 2))
            # CONSISTENCY_MARKER and unknown tokens are skipped (batch mode ignores them)
    return {k: np.array(v) for k, v in out.items()}


def save_se2_graph(path, poses, edges, info):
    """poses: (n, 3) x y theta; edges: (m, 5) i j dx dy dtheta; info: (m, 3, 3)"""
    with open(path, "w") as f:
        for i, p in enumerate(poses):
            f.write("VERTEX_SE2 %d %.17g %.17g %.17g\n" % (i, p[0], p[1], p[2]))
        iu = np.triu_indices(3)
        for e, m in zip(edges, info):
            f.write("EDGE_SE2 %d %d %.17g %.17g %.17g " % (int(e[0]), int(e[1]), e[2], e[3], e[4]))
            f.write(" ".join("%.17g" % x for x in m[iu]) + "\n")


def save_se3_graph(path, edges, info, poses=None):
    """3D pose graph in the reference's text format. edges: (m, 8) i j tx ty tz ax ay az (relative pose,
    rotation as axis-angle) written as `EDGE3:AXISANGLE` (ParsePrimitives.h:556-617: the measurement is taken
    as it stands, no roll-pitch-yaw conversion), info: (m, 6, 6) -> its 21 upper-triangular values row by row.
    poses (n, 6) [t | roll pitch yaw] are optional `VERTEX3` lines (:741-797, RPY as the parser expects);
    without them the reference initializes every pose by composing the edges, as it does for sphere2500."""
    iu = np.triu_indices(6)
    with open(path, "w") as f:
        if poses is not None:
            for i, p in enumerate(poses):
                f.write("VERTEX3 %d " % i + " ".join("%.17g" % x for x in p) + "\n")
        for e, m in zip(edges, info):
            f.write("EDGE3:AXISANGLE %d %d " % (int(e[0]), int(e[1])) + " ".join("%.17g" % x for x in e[2:8]) + " ")
            f.write(" ".join("%.17g" % x for x in np.asarray(m)[iu]) + "\n")


def save_ba_graph(path, cams, intr, points, obs, info=None, cam_id=None, pt_id=None):
    """Bundle adjustment graph in the reference's text format (data/Readme.txt, ParsePrimitives.h:861-931,
    805-850, 1123-1184): `VERTEX_CAM id x y z qx qy qz qw fx fy cx cy d` stores the camera-to-world pose (centre
    + quaternion), which the parser inverts into the world-to-camera [R | t] it optimizes; `VERTEX_XYZ id x y z`;
    `EDGE_PROJECT_P2MC point-id cam-id u v xx xy yy`.
      cams (nc, 6) world-to-camera [t | axis-angle] (the reference's internal CVertexCam state), intr (nc, 5)
      fx fy cx cy d, points (np, 3), obs (no, 4) point index, camera index, u, v; info (no, 2, 2) or None (identity);
      cam_id / pt_id: vertex ids (default: cameras 0..nc-1, points nc..nc+np-1)."""
    from scipy.spatial.transform import Rotation
    cams, intr, points, obs = (np.asarray(a, dtype=np.float64) for a in (cams, intr, points, obs))
    nc, npts = cams.shape[0], points.shape[0]
    cam_id = np.arange(nc) if cam_id is None else np.asarray(cam_id)
    pt_id = nc + np.arange(npts) if pt_id is None else np.asarray(pt_id)
    R = Rotation.from_rotvec(cams[:, 3:6])
    C = -R.inv().apply(cams[:, :3])            # camera centre in the world
    q = R.inv().as_quat()                       # x y z w, camera-to-world
    order = np.argsort(np.concatenate([cam_id, pt_id]), kind="stable")
    with open(path, "w") as f:
        for v in order:                         # vertices in id order, as the incremental datasets have them
            if v < nc:
                f.write("VERTEX_CAM %d " % cam_id[v] + " ".join("%.17g" % x for x in (*C[v], *q[v], *intr[v])) + "\n")
            else:
                f.write("VERTEX_XYZ %d " % pt_id[v - nc] + " ".join("%.17g" % x for x in points[v - nc]) + "\n")
        for k, o in enumerate(obs):
            m = np.eye(2) if info is None else np.asarray(info[k])
            f.write("EDGE_PROJECT_P2MC %d %d %.17g %.17g %.17g %.17g %.17g\n" % (
                pt_id[int(o[0])], cam_id[int(o[1])], o[2], o[3], m[0, 0], m[0, 1], m[1, 1]))


def load_bal(path):
    """Bundle Adjustment in the Large problem file: `ncams npoints nobs`, nobs lines `cam point x y`, then 9
    numbers per camera (Rodrigues vector, translation, f, k1, k2) and 3 per point. Returns dict(cam_index,
    point_index, xy (nobs, 2), cameras (ncams, 9), points (npoints, 3))."""
    with open(path) as f:
        tok = f.read().split()
    nc, npts, no = int(tok[0]), int(tok[1]), int(tok[2])
    o = np.array(tok[3:3 + 4 * no], dtype=np.float64).reshape(no, 4)
    rest = np.array(tok[3 + 4 * no:3 + 4 * no + 9 * nc + 3 * npts], dtype=np.float64)
    if rest.size != 9 * nc + 3 * npts:
        raise ValueError("truncated BAL file: %s" % path)
    return dict(cam_index=o[:, 0].astype(np.int64), point_index=o[:, 1].astype(np.int64), xy=o[:, 2:4].copy(),
                cameras=rest[:9 * nc].reshape(nc, 9), points=rest[9 * nc:].reshape(npts, 3))


def bal_to_slampp(bal):
    """BAL camera (9 parameters) -> the reference's 6 + 5. BAL projects p = -P / P.z with
    P = R X + t, then f (1 + k1 |p|^2 + k2 |p|^4) p: the camera looks down -z. The reference
    (BASolverBase.h:256-330) projects u = fx x / z + cx with x = R' X + t', then c + (1 + r^2 k) (u - c), r in
    PIXELS and k = d / ((fx + fy) / 2). With F = diag(1, -1, -1): R' = F R, t' = F t, observations (x, -y),
    fx = fy = f, cx = cy = 0; r = f |p| gives k = k1 / f^2, i.e. d = k1 / f. k2 has no counterpart and is dropped.
    The reference's own converter is not in its tree (data/Readme.txt points to an external script), so this
    mapping is derived from the two published camera models and checked by reprojection (tests/test_formats.py).
    Returns (cams (nc, 6) [t' | axis-angle of R'], intr (nc, 5), points, obs (no, 4) point, camera, u, v)."""
    from scipy.spatial.transform import Rotation
    cam = bal["cameras"]
    F = np.diag([1.0, -1.0, -1.0])
    R = Rotation.from_rotvec(cam[:, :3]).as_matrix()
    Rp = np.einsum("ij,cjk->cik", F, R)
    cams = np.concatenate([cam[:, 3:6] @ F.T, Rotation.from_matrix(Rp).as_rotvec()], axis=1)
    f = cam[:, 6]
    intr = np.stack([f, f, np.zeros_like(f), np.zeros_like(f), cam[:, 7] / f], axis=1)
    obs = np.stack([bal["point_index"].astype(np.float64), bal["cam_index"].astype(np.float64),
                    bal["xy"][:, 0], -bal["xy"][:, 1]], axis=1)
    return cams, intr, bal["points"].copy(), obs


def convert_bal_file(path_bal, path_graph):
    """BAL problem file -> VERTEX_CAM / VERTEX_XYZ / EDGE_PROJECT_P2MC graph the reference's parser reads."""
    cams, intr, points, obs = bal_to_slampp(load_bal(path_bal))
    save_ba_graph(path_graph, cams, intr, points, obs)


def se2_linearize(poses, edges, info):
    """Hot-path inputs (synth.Problem) of a 2D pose graph at the given estimate: analytic Jacobians of
    the relative-pose error (the quantity reference include/slam/2DSolverBase.h:269-373 computes),
    residual = measurement - prediction with the angle wrapped to (-pi, pi]."""
    from .synth import Problem
    poses = np.asarray(poses, dtype=np.float64)
    edges = np.asarray(edges, dtype=np.float64)
    v0 = edges[:, 0].astype(np.int64)
    v1 = edges[:, 1].astype(np.int64)
    ne = v0.size
    c, s = np.cos(poses[v0, 2]), np.sin(poses[v0, 2])
    d = poses[v1, :2] - poses[v0, :2]
    J0 = np.zeros((ne, 3, 3))
    J1 = np.zeros((ne, 3, 3))
    J0[:, 0, 0], J0[:, 0, 1], J0[:, 0, 2] = -c, -s, -s * d[:, 0] + c * d[:, 1]
    J0[:, 1, 0], J0[:, 1, 1], J0[:, 1, 2] = s, -c, -c * d[:, 0] - s * d[:, 1]
    J0[:, 2, 2] = -1
    J1[:, 0, 0], J1[:, 0, 1] = c, s
    J1[:, 1, 0], J1[:, 1, 1] = -s, c
    J1[:, 2, 2] = 1
    pred = np.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1], poses[v1, 2] - poses[v0, 2]], axis=1)
    r = edges[:, 2:5] - pred
    r[:, 2] = (r[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return Problem(name="se2_graph", dim=np.full(poses.shape[0], 3, dtype=np.int32), v0=v0, v1=v1, d0=3, d1=3, rd=3,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(ne, 9),
                   J1=np.ascontiguousarray(J1.transpose(0, 2, 1)).reshape(ne, 9),
                   Om=np.asarray(info, dtype=np.float64).reshape(ne, 9), r=r, unary_vertex=0, damping=0.0)


def _hat(v):
    z = np.zeros(v.shape[0])
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1),
                     np.stack([-v[:, 1], v[:, 0], z], 1)], 1)


def se3_linearize(poses, edges, info):
    """Hot-path inputs (synth.Problem) of a 3D pose graph at the given estimate. poses (n, 6) [t | axis-angle],
    edges (m, 8) i j + 6D measurement, info (m, 6, 6). Expectation C3DJacobians::Absolute_to_Relative, error
    of CEdgePose3D (include/slam/SE3_Types.h:264-286); Jacobians analytic w.r.t. the increments of
    Relative_to_Absolute (the reference takes forward differences there, 3DSolverBase.h:1331-1371)."""
    from scipy.spatial.transform import Rotation
    from .synth import Problem
    poses = np.asarray(poses, dtype=np.float64)
    edges = np.asarray(edges, dtype=np.float64)
    v0, v1 = edges[:, 0].astype(np.int64), edges[:, 1].astype(np.int64)
    ne = v0.size
    R1 = Rotation.from_rotvec(poses[v0, 3:]).as_matrix()
    R2 = Rotation.from_rotvec(poses[v1, 3:]).as_matrix()
    et = np.einsum("eji,ej->ei", R1, poses[v1, :3] - poses[v0, :3])
    Re = np.einsum("eji,ejk->eik", R1, R2)
    er = Rotation.from_matrix(Re).as_rotvec()
    th = np.linalg.norm(er, axis=1)
    small = th < 1e-4
    ths = np.where(small, 1.0, th)
    c = np.where(small, 1.0 / 12 + th ** 2 / 720, 1 / ths ** 2 - (1 + np.cos(ths)) / (2 * ths * np.sin(ths)))
    K = _hat(er)
    Ji = np.eye(3)[None] + 0.5 * K + c[:, None, None] * np.einsum("eij,ejk->eik", K, K)
    J0 = np.zeros((ne, 6, 6))
    J1 = np.zeros((ne, 6, 6))
    J0[:, :3, :3] = -np.eye(3)
    J0[:, :3, 3:] = _hat(et)
    J0[:, 3:, 3:] = -np.einsum("eij,ekj->eik", Ji, Re)
    J1[:, :3, :3] = Re
    J1[:, 3:, 3:] = Ji
    z = edges[:, 2:8]
    rrot = (Rotation.from_rotvec(z[:, 3:]) * Rotation.from_matrix(Re).inv()).as_rotvec()
    r = np.concatenate([z[:, :3] - et, rrot], axis=1)
    return Problem(name="se3_graph", dim=np.full(poses.shape[0], 6, dtype=np.int32), v0=v0, v1=v1, d0=6, d1=6, rd=6,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(ne, 36),
                   J1=np.ascontiguousarray(J1.transpose(0, 2, 1)).reshape(ne, 36),
                   Om=np.asarray(info, dtype=np.float64).reshape(ne, 36), r=r, unary_vertex=0, damping=0.0)


def se3_plus(poses, dx):
    """x (+) dx of CVertexPose3D (C3DJacobians::Relative_to_Absolute): t + R dt, R exp(dr)"""
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(poses[:, 3:])
    out = np.empty_like(poses)
    out[:, :3] = poses[:, :3] + R.apply(dx[:, :3])
    out[:, 3:] = (R * Rotation.from_rotvec(dx[:, 3:])).as_rotvec()
    return out


def ba_linearize(cams, intr, points, obs, cam_id=None, pt_id=None):
    """Hot-path inputs (synth.Problem) of a bundle adjustment at the given estimate, reference parameterization:
    cams (nc, 6) [t | axis-angle] world -> camera, intr (nc, 5) fx fy cx cy k, points (np, 3), obs (no, 4)
    cam pt u v. Model and increments of CBAJacobians::Project_P2C (include/slam/BASolverBase.h:260-325,559-620;
    analytic where the reference takes forward differences). Vertex ids: cameras 0..nc-1, points nc.. unless
    cam_id / pt_id say otherwise."""
    from scipy.spatial.transform import Rotation
    from .synth import Problem
    cams, intr, points, obs = (np.asarray(a, dtype=np.float64) for a in (cams, intr, points, obs))
    nc, npts, no = cams.shape[0], points.shape[0], obs.shape[0]
    co, po = obs[:, 0].astype(np.int64), obs[:, 1].astype(np.int64)
    cam_id = np.arange(nc) if cam_id is None else np.asarray(cam_id)
    pt_id = nc + np.arange(npts) if pt_id is None else np.asarray(pt_id)
    R = Rotation.from_rotvec(cams[:, 3:]).as_matrix()[co]
    X = points[po]
    x = np.einsum("eij,ej->ei", R, X) + cams[co, :3]
    fx, fy, cx, cy = (intr[co, i] for i in range(4))
    k = intr[co, 4] / (0.5 * (fx + fy))
    iz = 1.0 / x[:, 2]
    d = np.stack([fx * x[:, 0] * iz, fy * x[:, 1] * iz], axis=1)
    r2 = (d ** 2).sum(axis=1)
    g = 1 + r2 * k
    uv = np.stack([cx, cy], axis=1) + g[:, None] * d
    Jd = np.zeros((no, 2, 3))
    Jd[:, 0, 0], Jd[:, 0, 2] = fx * iz, -fx * x[:, 0] * iz * iz
    Jd[:, 1, 1], Jd[:, 1, 2] = fy * iz, -fy * x[:, 1] * iz * iz
    D = g[:, None, None] * np.eye(2)[None] + 2 * k[:, None, None] * np.einsum("ei,ej->eij", d, d)
    PR = np.einsum("eij,ejk,ekl->eil", D, Jd, R)
    J0 = np.concatenate([PR, -np.einsum("eij,ejk->eik", PR, _hat(X))], axis=2)
    dim = np.empty(nc + npts, dtype=np.int32)
    dim[cam_id] = 6
    dim[pt_id] = 3
    return Problem(name="ba", dim=dim, v0=cam_id[co], v1=pt_id[po], d0=6, d1=3, rd=2,
                   J0=np.ascontiguousarray(J0.transpose(0, 2, 1)).reshape(no, 12),
                   J1=np.ascontiguousarray(PR.transpose(0, 2, 1)).reshape(no, 6),
                   Om=np.tile(np.eye(2).ravel(), (no, 1)), r=obs[:, 2:4] - uv, unary_vertex=0, damping=0.0)


def problem_from_graph(path):
    """Graph file -> hot-path inputs (synth.Problem) at the file's initial estimate, plus a short description.
    2D / 3D pose graphs without VERTEX lines are initialized the way the reference's parse loop does it
    (CEdgePose2D / CEdgePose3D constructors: an unseen second vertex becomes first (+) measurement, in file order);
    BA files use the reference's VERTEX_CAM convention (camera-to-world in the file)."""
    from scipy.spatial.transform import Rotation
    g = load_graph(path)
    if g["projections"].size:
        cams_f, pts_f, proj = g["cams"], g["points"], g["projections"]
        q = Rotation.from_quat(cams_f[:, 4:8]).inv()          # the parser's inversion (ParsePrimitives.h:886-905)
        cams = np.concatenate([q.apply(-cams_f[:, 1:4]), q.as_rotvec()], axis=1)
        ids = np.concatenate([cams_f[:, 0], pts_f[:, 0]]).astype(np.int64)
        nv = int(ids.max()) + 1
        cam_id, pt_id = cams_f[:, 0].astype(np.int64), pts_f[:, 0].astype(np.int64)
        cam_index = np.full(nv, -1, dtype=np.int64)
        cam_index[cam_id] = np.arange(cam_id.size)
        pt_index = np.full(nv, -1, dtype=np.int64)
        pt_index[pt_id] = np.arange(pt_id.size)
        obs = np.stack([cam_index[proj[:, 1].astype(np.int64)].astype(np.float64),     # ba_linearize: cam pt u v
                        pt_index[proj[:, 0].astype(np.int64)].astype(np.float64), proj[:, 2], proj[:, 3]], axis=1)
        prob = ba_linearize(cams, cams_f[:, 8:13], pts_f[:, 1:4], obs, cam_id=cam_id, pt_id=pt_id)
        prob["Om"] = np.ascontiguousarray(g["proj_info"]).reshape(-1, 4)
        prob["nc"], prob["npts"] = cam_id.size, pt_id.size
        return prob, "BA graph file (%d cameras, %d points, %d observations)" % (cam_id.size, pt_id.size, obs.shape[0])
    if g["se3_edges"].size:
        e = g["se3_edges"]
        n = int(e[:, :2].max()) + 1
        poses = np.zeros((n, 6))
        seen = np.zeros(n, dtype=bool)
        for v in g["se3_vertices"].reshape(-1, 7):  # VERTEX3: t + roll pitch yaw (ParsePrimitives.h:741-797)
            poses[int(v[0])] = np.concatenate([v[1:4], Rotation.from_euler("ZYX", [v[6], v[5], v[4]]).as_rotvec()])
            seen[int(v[0])] = True
        if not seen.any():
            seen[int(e[0, 0])] = True
        for a, b, *z in e:
            a, b = int(a), int(b)
            if seen[a] and not seen[b]:
                poses[b] = se3_plus(poses[a:a + 1], np.asarray(z)[None, :])[0]
                seen[b] = True
        if not seen.all():
            raise ValueError("3D pose graph: %d poses are not reachable through forward edges" % int((~seen).sum()))
        return se3_linearize(poses, e, g["se3_info"]), "3D pose graph file (%d poses, %d edges)" % (n, e.shape[0])
    if g["se2_edges"].size:
        e = g["se2_edges"]
        n = int(e[:, :2].max()) + 1
        poses = np.zeros((n, 3))
        seen = np.zeros(n, dtype=bool)
        for v in g["se2_vertices"].reshape(-1, 4):
            poses[int(v[0])] = v[1:4]
            seen[int(v[0])] = True
        if not seen.any():
            seen[int(e[0, 0])] = True
        for a, b, dx, dy, dt in e:
            a, b = int(a), int(b)
            if seen[a] and not seen[b]:
                c, s = np.cos(poses[a, 2]), np.sin(poses[a, 2])
                poses[b] = [poses[a, 0] + c * dx - s * dy, poses[a, 1] + s * dx + c * dy, poses[a, 2] + dt]
                seen[b] = True
        if not seen.all():
            raise ValueError("2D pose graph: %d poses are not reachable through forward edges" % int((~seen).sum()))
        return se2_linearize(poses, e, g["se2_info"]), "2D pose graph file (%d poses, %d edges)" % (n, e.shape[0])
    raise ValueError("no edges of a known type in %s" % path)
