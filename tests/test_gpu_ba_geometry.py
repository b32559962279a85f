"""On-device BA edge geometry (spp_ba_linearize_device / spp_ba_update_device) against golden vectors
produced by the REFERENCE itself (tests/golden/ba_geometry.npz, tools/make_golden_bageom.py:
CBAJacobians::Project_P2C with its Jacobians, include/slam/BASolverBase.h:260-325,559-620, and the SE(3)
composition C3DJacobians::Relative_to_Absolute, include/slam/3DSolverBase.h:807-850).

Tolerances: the expectation is closed-form on both sides -> 1e-11 px. The reference's Jacobians are forward
differences with delta = 1e-9 of pixel values ~1e2..1e3, i.e. they carry ~1e-4 absolute / 1e-7 relative noise
themselves; the device Jacobians are analytic, and |J_gpu - J_ref| <= 1e-3 + 1e-6 |J_ref| is demanded."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import api

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ba_geometry.npz"))


def test_projection_error_and_jacobians_match_the_reference():
    n = G["cam"].shape[0]
    ctx = api.Context(0)
    idx = np.arange(n, dtype=np.int32)
    meas = G["uv"] + np.array([0.25, -0.5])   # r = z - uv must come out as this offset
    d = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(v).ravel()) for k, v in
         dict(cam_of=idx, pt_of=idx, cams=G["cam"], intr=G["intr"], pts=G["X"], meas=meas).items()}
    J0, J1, r = api.DeviceArray(ctx, 12 * n), api.DeviceArray(ctx, 6 * n), api.DeviceArray(ctx, 2 * n)
    ctx.ba_linearize_device(n, d["cam_of"].ptr, d["pt_of"].ptr, d["cams"].ptr, d["intr"].ptr, d["pts"].ptr, d["meas"].ptr,
                            J0.ptr, J1.ptr, r.ptr)
    ctx.synchronize()
    rr = r.download().reshape(n, 2)
    assert np.abs(rr - np.array([0.25, -0.5])).max() < 1e-11
    for got, ref in ((J0.download().reshape(n, 12), G["H1"]), (J1.download().reshape(n, 6), G["H2"])):
        assert np.all(np.abs(got - ref) <= 1e-3 + 1e-6 * np.abs(ref)), np.abs(got - ref).max()
    # gathered indices: a permuted observation list gives the permuted rows
    perm = np.random.default_rng(0).permutation(n).astype(np.int32)
    dperm = api.DeviceArray.from_host(ctx, perm)
    dmeas = api.DeviceArray.from_host(ctx, meas[perm].ravel())
    J0b = api.DeviceArray(ctx, 12 * n)
    ctx.ba_linearize_device(n, dperm.ptr, dperm.ptr, d["cams"].ptr, d["intr"].ptr, d["pts"].ptr, dmeas.ptr, J0b.ptr, J1.ptr, r.ptr)
    ctx.synchronize()
    assert np.array_equal(J0b.download().reshape(n, 12), J0.download().reshape(n, 12)[perm])
    ctx.close()


def test_camera_and_point_update_match_the_reference_composition():
    n = G["cam"].shape[0]
    ctx = api.Context(0)
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(n, 3))
    dpt = rng.normal(size=(n, 3))
    # dx laid out like a Lambda with interleaved vertices: [cam 0 | point 0 | cam 1 | point 1 ...]
    dx = np.zeros(9 * n)
    cam_off = 9 * np.arange(n, dtype=np.int64)
    pt_off = cam_off + 6
    for i in range(n):
        dx[cam_off[i]:cam_off[i] + 6] = G["inc"][i]
        dx[pt_off[i]:pt_off[i] + 3] = dpt[i]
    dc, dp = api.DeviceArray.from_host(ctx, G["cam"].ravel()), api.DeviceArray.from_host(ctx, pts.ravel())
    dco, dpo = api.DeviceArray.from_host(ctx, cam_off), api.DeviceArray.from_host(ctx, pt_off)
    ddx = api.DeviceArray.from_host(ctx, dx)
    nrm = ctx.ba_update_device(n, dc.ptr, dco.ptr, n, dp.ptr, dpo.ptr, ddx.ptr, dx.size, apply=False)
    assert abs(nrm - np.linalg.norm(dx)) < 1e-12 * np.linalg.norm(dx)
    assert np.array_equal(dc.download(), G["cam"].ravel())
    ctx.ba_update_device(n, dc.ptr, dco.ptr, n, dp.ptr, dpo.ptr, ddx.ptr, dx.size, apply=True)
    assert np.abs(dc.download().reshape(n, 6) - G["composed"]).max() < 1e-12
    assert np.array_equal(dp.download().reshape(n, 3), pts + dpt)
    ctx.close()


@pytest.mark.parametrize("name", ["ba_small", "ba_interleaved", "ladybug49"])
def test_resident_ba_iteration_reduces_the_reprojection_error(name):
    """One damped Gauss-Newton (= LM with fixed damping) iteration entirely in HBM: device linearization
    (reference parameterization) -> device assembly -> device solve -> device (+). The linearized residual
    reproduces the problem's, the sum of squared reprojection errors drops, and the solve agrees with the
    CPU oracle on the device-linearized system."""
    from slam_plus_plus_amd import synth
    from oracle import spp_oracle as orc
    prob = synth.make(name)
    s = synth.ba_states(prob)
    no, nc, npts = prob.v0.size, s["cams"].shape[0], s["points"].shape[0]
    ctx = api.Context(0)
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, 6, 3, 2, prob.unary_vertex)
    d = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(v).ravel()) for k, v in s.items()}
    dOm = api.DeviceArray.from_host(ctx, prob.Om.ravel())
    J0, J1, r = api.DeviceArray(ctx, 12 * no), api.DeviceArray(ctx, 6 * no), api.DeviceArray(ctx, 2 * no)
    dv, de = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n)

    def linearize():
        ctx.ba_linearize_device(no, d["cam_of"].ptr, d["pt_of"].ptr, d["cams"].ptr, d["intr"].ptr, d["points"].ptr,
                                d["meas"].ptr, J0.ptr, J1.ptr, r.ptr)
        ctx.synchronize()
        return r.download().reshape(no, 2)

    r0 = linearize()
    assert np.abs(r0 - prob.r).max() < 1e-9          # z was built as projection + prob.r
    # LM damping relative to THIS parameterization's Hessian diagonal (NonlinearSolver_Lambda_LM.h:151-199)
    j0 = J0.download().reshape(no, 6, 2)
    damping = 1e-3 * float((j0 ** 2).sum(axis=2).max())
    ctx.assemble_device(J0.ptr, J1.ptr, dOm.ptr, r.ptr, damping, dv.ptr, de.ptr)
    lam, eta = st.with_vals(dv.download()), de.download()
    ctx.analyze(st, api.MODE_AUTO)
    assert ctx.factor_solve_device(dv.ptr, de.ptr) == 0
    dx = de.download()
    code, xo, _ = orc.schur_solve(lam, eta)
    assert code == 0 and np.linalg.norm(dx - xo) / np.linalg.norm(xo) < 1e-10
    nrm = ctx.ba_update_device(nc, d["cams"].ptr, d["cam_dxoff"].ptr, npts, d["points"].ptr, d["pt_dxoff"].ptr,
                               de.ptr, st.n, apply=True)
    assert abs(nrm - np.linalg.norm(dx)) <= 1e-12 * np.linalg.norm(dx)
    r1 = linearize()
    assert (r1 ** 2).sum() < 0.9 * (r0 ** 2).sum(), ((r0 ** 2).sum(), (r1 ** 2).sum())
    ctx.close()
