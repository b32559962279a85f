"""On-device SE(3) pose-pose edge geometry (spp_se3_linearize_device / spp_se3_update_device) against golden
vectors produced by the REFERENCE itself (tests/golden/se3_geometry.npz, tools/make_golden_se3geom.py):
C3DJacobians::Absolute_to_Relative + forward-difference Jacobians (include/slam/3DSolverBase.h:1331-1371), the
CEdgePose3D error (include/slam/SE3_Types.h:264-286) and the vertex (+) (3DSolverBase.h:807-850).

Error and composition are closed-form on both sides: 1e-12. The reference's Jacobians are forward differences
with delta = 1e-9 (noise ~1e-6 absolute on entries of magnitude 1..50); the device ones are analytic:
|J_gpu - J_ref| <= 1e-5 + 1e-6 |J_ref|."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import api

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "se3_geometry.npz"))


def test_error_and_jacobians_match_the_reference():
    n = G["v1"].shape[0]
    ctx = api.Context(0)
    poses = np.concatenate([G["v1"], G["v2"]], axis=0)          # vertices 0..n-1 = v1, n..2n-1 = v2
    i0 = np.arange(n, dtype=np.int32)
    d = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(v).ravel()) for k, v in
         dict(v0=i0, v1=i0 + n, poses=poses, meas=G["z"]).items()}
    J0, J1, r = api.DeviceArray(ctx, 36 * n), api.DeviceArray(ctx, 36 * n), api.DeviceArray(ctx, 6 * n)
    ctx.se3_linearize_device(n, d["v0"].ptr, d["v1"].ptr, d["poses"].ptr, d["meas"].ptr, J0.ptr, J1.ptr, r.ptr)
    ctx.synchronize()
    assert np.abs(r.download().reshape(n, 6) - G["err"]).max() < 1e-12
    for got, ref in ((J0.download().reshape(n, 36), G["H1"]), (J1.download().reshape(n, 36), G["H2"])):
        assert np.all(np.abs(got - ref) <= 1e-5 + 1e-6 * np.abs(ref)), np.abs(got - ref).max()
    ctx.close()


def test_pose_update_matches_the_reference_composition():
    n = G["v1"].shape[0]
    ctx = api.Context(0)
    dp = api.DeviceArray.from_host(ctx, G["v1"].ravel())
    dd = api.DeviceArray.from_host(ctx, G["inc"].ravel())
    nrm = ctx.se3_update_device(n, dp.ptr, dd.ptr, apply=False)
    assert abs(nrm - np.linalg.norm(G["inc"])) < 1e-12 * np.linalg.norm(G["inc"])
    assert np.array_equal(dp.download(), G["v1"].ravel())
    ctx.se3_update_device(n, dp.ptr, dd.ptr, apply=True)
    assert np.abs(dp.download().reshape(n, 6) - G["composed"]).max() < 1e-12
    ctx.close()
