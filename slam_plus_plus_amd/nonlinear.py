"""Batch Gauss-Newton over the Lambda-solve hot path: the loop glue of the reference's
CNonlinearSolver_Lambda::Optimize (include/slam/NonlinearSolver_Lambda.h:539-666)
for 2D and 3D pose graphs. Product paths: `_ResidentPath` (the whole iteration in HBM: device
linearization, assembly, solve, ||dx||, vertex update) and `_DevicePath` (Jacobians on the host --
SURVEY 8d metric 2's baseline wording -- assembly and solve on the device).

Per iteration, exactly in the reference's order (:605-664):
    linearize at the current estimate  ->  Lambda = J^T Omega J (+ unary factor), eta = J^T Omega r
    dx = Lambda^-1 eta            (symbolic analysis once: the structure is fixed within Optimize)
    if ||dx|| <= f_min_dx_norm: stop WITHOUT applying dx
    x <- x (+) dx                 (CVertexPose2D::Operator_Plus, SE2_Types.h:70-74: add, clamp the angle)
    if the factorization failed: stop, estimate unchanged
Defaults are slam_app's: 5 iterations, threshold 0.01 (src/slam_app/Main.cpp:706-707).
"""
import math

import numpy as np

from . import api
from .formats import se2_linearize, se3_linearize, se3_plus, ba_linearize


class CPoseGraph2D:
    """The 'system': vertex states (n, 3) x y theta, edges (m, 5) i j dx dy dtheta, information (m, 3, 3)."""

    def __init__(self, poses, edges, info):
        self.poses = np.array(poses, dtype=np.float64)
        self.edges = np.asarray(edges, dtype=np.float64)
        self.info = np.asarray(info, dtype=np.float64)

    dof = 3

    def linearize(self):
        return se2_linearize(self.poses, self.edges, self.info)

    def plus(self, dx):
        self.poses += dx.reshape(-1, 3)
        self.poses[:, 2] = np.fmod(self.poses[:, 2], 2 * math.pi)  # f_ClampAngle_2Pi, 2DSolverBase.h:44

    def chi2(self):
        prob = self.linearize()
        om = prob.Om.reshape(-1, 3, 3)
        return float(np.einsum("ei,eij,ej->", prob.r, om, prob.r))


class CPoseGraph3D:
    """3D pose graph: vertex states (n, 6) [t | axis-angle], edges (m, 8) i j + 6D measurement, information (m, 6, 6)
    (CVertexPose3D / CEdgePose3D, include/slam/SE3_Types.h)"""
    dof = 6

    def __init__(self, poses, edges, info):
        self.poses = np.array(poses, dtype=np.float64)
        self.edges = np.asarray(edges, dtype=np.float64)
        self.info = np.asarray(info, dtype=np.float64)

    def linearize(self):
        return se3_linearize(self.poses, self.edges, self.info)

    def plus(self, dx):
        self.poses = se3_plus(self.poses, dx.reshape(-1, 6))

    def chi2(self):
        prob = self.linearize()
        return float(np.einsum("ei,eij,ej->", prob.r, prob.Om.reshape(-1, 6, 6), prob.r))


class _DevicePath:
    """device assembly + device solve through the C ABI (the product path; needs the GPU)"""

    def __init__(self, device=0):
        self.ctx = api.Context(device)
        self.st = None

    def solve(self, prob, first):
        ctx = self.ctx
        if first:
            self.st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
            self.d_vals = api.DeviceArray(ctx, self.st.nvals)
            self.d_eta = api.DeviceArray(ctx, self.st.n)
            self.d_in = [api.DeviceArray(ctx, a.size) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
        for d, a in zip(self.d_in, (prob.J0, prob.J1, prob.Om, prob.r)):
            d.upload(np.ascontiguousarray(a).ravel())
        ctx.assemble_device(self.d_in[0].ptr, self.d_in[1].ptr, self.d_in[2].ptr, self.d_in[3].ptr, prob.damping,
                            self.d_vals.ptr, self.d_eta.ptr)
        if first:
            ctx.analyze(self.st, api.MODE_AUTO)  # FinalBlockStructure + symbolic: once per Optimize
        code = ctx.factor_solve_device(self.d_vals.ptr, self.d_eta.ptr)
        if code != 0:
            return False, None
        return True, self.d_eta.download()

    def close(self):
        self.ctx.close()


class _ResidentPath:
    """the whole Gauss-Newton iteration in HBM: linearization (spp_se2_linearize_device), assembly, solve,
    ||dx|| and the vertex update (spp_se2_update_device). Host traffic per iteration: 8 bytes (the norm)."""

    def __init__(self, device=0):
        self.ctx = api.Context(device)

    def begin(self, system):
        ctx = self.ctx
        prob = system.linearize()   # only for the (constant) structure + Omega
        self.nv, self.ne, self.dof = system.poses.shape[0], system.edges.shape[0], system.dof
        self.st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, self.dof, self.dof, self.dof, prob.unary_vertex)
        self._lin = ctx.se2_linearize_device if self.dof == 3 else ctx.se3_linearize_device
        self._upd = ctx.se2_update_device if self.dof == 3 else ctx.se3_update_device
        self.d_v0 = api.DeviceArray.from_host(ctx, prob.v0.astype(np.int32))
        self.d_v1 = api.DeviceArray.from_host(ctx, prob.v1.astype(np.int32))
        self.d_meas = api.DeviceArray.from_host(ctx, np.ascontiguousarray(system.edges[:, 2:2 + self.dof]).ravel())
        self.d_poses = api.DeviceArray.from_host(ctx, system.poses.ravel())
        self.d_Om = api.DeviceArray.from_host(ctx, np.ascontiguousarray(prob.Om).ravel())
        dd = self.dof * self.dof
        self.d_J0, self.d_J1 = api.DeviceArray(ctx, dd * self.ne), api.DeviceArray(ctx, dd * self.ne)
        self.d_r = api.DeviceArray(ctx, self.dof * self.ne)
        self.d_vals, self.d_eta = api.DeviceArray(ctx, self.st.nvals), api.DeviceArray(ctx, self.st.n)
        self.analyzed = False

    def step(self):
        """linearize + assemble + solve; returns (ok, ||dx||); dx stays on the device"""
        ctx = self.ctx
        self._lin(self.ne, self.d_v0.ptr, self.d_v1.ptr, self.d_poses.ptr, self.d_meas.ptr,
                  self.d_J0.ptr, self.d_J1.ptr, self.d_r.ptr)
        ctx.assemble_device(self.d_J0.ptr, self.d_J1.ptr, self.d_Om.ptr, self.d_r.ptr, 0.0, self.d_vals.ptr, self.d_eta.ptr)
        if not self.analyzed:
            ctx.analyze(self.st, api.MODE_AUTO)
            self.analyzed = True
        if ctx.factor_solve_device(self.d_vals.ptr, self.d_eta.ptr) != 0:
            return False, 0.0
        return True, self._upd(self.nv, self.d_poses.ptr, self.d_eta.ptr, apply=False)

    def apply(self):
        self._upd(self.nv, self.d_poses.ptr, self.d_eta.ptr, apply=True)

    def finish(self, system):
        system.poses[:] = self.d_poses.download().reshape(-1, self.dof)

    def close(self):
        self.ctx.close()


class CNonlinearSolver_Lambda:
    """mirror of the reference class for CPoseGraph2D systems. `path` may be replaced by any object with
    solve(problem, first) -> (ok, dx) (the tests drive the loop glue with a CPU checker that way)."""

    def __init__(self, system, path=None, device=0, verbose=False, host_jacobians=False):
        self.system = system
        if path is None:  # the product paths need the GPU; host_jacobians keeps the linearization in numpy
            path = _DevicePath(device) if host_jacobians else _ResidentPath(device)
        self.path = path
        self.verbose = verbose
        self.n_iterations = 0
        self.last_dx_norm = None

    def Optimize(self, n_max_iteration_num=5, f_min_dx_norm=0.01):
        s = self.system
        self.n_iterations = 0
        if hasattr(self.path, "step"):  # device-resident iteration
            self.path.begin(s)
            for it in range(n_max_iteration_num):
                ok, norm = self.path.step()
                self.n_iterations = it + 1
                self.last_dx_norm = norm
                if self.verbose:
                    print("%s, residual norm: %.4f" % ("Cholesky succeeded" if ok else "Cholesky failed", norm))
                if norm <= f_min_dx_norm or not ok:
                    break
                self.path.apply()
            self.path.finish(s)
            return self.n_iterations
        for it in range(n_max_iteration_num):
            prob = s.linearize()
            ok, dx = self.path.solve(prob, it == 0)
            self.n_iterations = it + 1
            norm = float(np.linalg.norm(dx)) if ok else 0.0
            self.last_dx_norm = norm
            if self.verbose:
                print("%s, residual norm: %.4f" % ("Cholesky succeeded" if ok else "Cholesky failed", norm))
            if norm <= f_min_dx_norm:
                break
            if ok:
                s.plus(dx)
            else:
                break
        return self.n_iterations


# --------------------------------------------------------------------------------------------------
# Levenberg-Marquardt (what slam_app silently uses for every BA input, src/slam_app/Main.cpp:203-208)
# --------------------------------------------------------------------------------------------------
class CBundleAdjustment:
    """BA 'system': cams (nc, 6) [t | axis-angle] world -> camera, intr (nc, 5), points (np, 3), obs (no, 4)
    cam pt u v. Vertex ids: cameras 0..nc-1, points nc.. (the layout of the reference's BA example)."""

    def __init__(self, cams, intr, points, obs):
        self.cams = np.array(cams, dtype=np.float64)
        self.intr = np.asarray(intr, dtype=np.float64)
        self.points = np.array(points, dtype=np.float64)
        self.obs = np.asarray(obs, dtype=np.float64)

    def linearize(self):
        return ba_linearize(self.cams, self.intr, self.points, self.obs)

    def chi2(self):
        r = self.linearize().r
        return float((r ** 2).sum())

    def state(self):
        return self.cams.copy(), self.points.copy()

    def set_state(self, st):
        self.cams, self.points = st[0].copy(), st[1].copy()

    def plus(self, dx):
        nc = self.cams.shape[0]
        self.cams = se3_plus(self.cams, dx[:6 * nc].reshape(nc, 6))
        self.points = self.points + dx[6 * nc:].reshape(-1, 3)


class _ResidentBAPath:
    """LM iteration pieces in HBM: spp_ba_linearize_device, spp_assemble_device (damping alpha),
    spp_factor_solve_device, spp_ba_update_device, chi2 / alpha0 / gain-ratio reductions."""

    def __init__(self, device=0):
        self.ctx = api.Context(device)

    def begin(self, system):
        ctx, s = self.ctx, system
        prob = s.linearize()   # structure only
        self.no, self.nc, self.np = s.obs.shape[0], s.cams.shape[0], s.points.shape[0]
        self.st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, 6, 3, 2, prob.unary_vertex)
        up = lambda a: api.DeviceArray.from_host(ctx, np.ascontiguousarray(a).ravel())
        self.d_cam_of, self.d_pt_of = up(s.obs[:, 0].astype(np.int32)), up(s.obs[:, 1].astype(np.int32))
        self.d_cams, self.d_intr, self.d_pts = up(s.cams), up(s.intr), up(s.points)
        self.d_meas, self.d_Om = up(s.obs[:, 2:4]), up(prob.Om)
        self.d_cam_off = up(6 * np.arange(self.nc, dtype=np.int64))
        self.d_pt_off = up(6 * self.nc + 3 * np.arange(self.np, dtype=np.int64))
        self.d_J0, self.d_J1 = api.DeviceArray(ctx, 12 * self.no), api.DeviceArray(ctx, 6 * self.no)
        self.d_r = api.DeviceArray(ctx, 2 * self.no)
        self.d_vals, self.d_eta, self.d_dx = (api.DeviceArray(ctx, self.st.nvals), api.DeviceArray(ctx, self.st.n),
                                              api.DeviceArray(ctx, self.st.n))
        self.s_cams, self.s_pts = api.DeviceArray(ctx, 6 * self.nc), api.DeviceArray(ctx, 3 * self.np)
        self.analyzed = False

    def linearize(self):
        self.ctx.ba_linearize_device(self.no, self.d_cam_of.ptr, self.d_pt_of.ptr, self.d_cams.ptr, self.d_intr.ptr,
                                     self.d_pts.ptr, self.d_meas.ptr, self.d_J0.ptr, self.d_J1.ptr, self.d_r.ptr)

    def chi2(self):
        """error at the CURRENT state (re-evaluates the residuals; J of the last linearization is overwritten too,
        which the loop accounts for by re-linearizing after a rejected step is rolled back)"""
        self.linearize()
        return self.ctx.edge_chi2_device(self.no, 2, self.d_r.ptr, self.d_Om.ptr)

    def max_hessian_diag(self):
        return self.ctx.edge_hessian_maxdiag_device(self.no, 2, 6, 3, self.d_J0.ptr, self.d_J1.ptr, self.d_Om.ptr)

    def solve(self, alpha):
        ctx = self.ctx
        ctx.assemble_device(self.d_J0.ptr, self.d_J1.ptr, self.d_Om.ptr, self.d_r.ptr, alpha, self.d_vals.ptr, self.d_eta.ptr)
        if not self.analyzed:
            ctx.analyze(self.st, api.MODE_AUTO)
            self.analyzed = True
        self.d_dx.copy_from(self.d_eta)
        if ctx.factor_solve_device(self.d_vals.ptr, self.d_dx.ptr) != 0:
            return False, 0.0
        return True, ctx.ba_update_device(self.nc, self.d_cams.ptr, self.d_cam_off.ptr, self.np, self.d_pts.ptr,
                                          self.d_pt_off.ptr, self.d_dx.ptr, self.st.n, apply=False)

    def gain_denominator(self, alpha):
        return self.ctx.lm_gain_denominator_device(self.st.n, self.d_dx.ptr, self.d_eta.ptr, alpha)

    def save(self):
        self.s_cams.copy_from(self.d_cams)
        self.s_pts.copy_from(self.d_pts)

    def restore(self):
        self.d_cams.copy_from(self.s_cams)
        self.d_pts.copy_from(self.s_pts)

    def apply(self):
        self.ctx.ba_update_device(self.nc, self.d_cams.ptr, self.d_cam_off.ptr, self.np, self.d_pts.ptr, self.d_pt_off.ptr,
                                  self.d_dx.ptr, self.st.n, apply=True)

    def finish(self, system):
        system.cams = self.d_cams.download().reshape(-1, 6)
        system.points = self.d_pts.download().reshape(-1, 3)

    def close(self):
        self.ctx.close()


class CNonlinearSolver_Lambda_LM:
    """Mirror of CNonlinearSolver_Lambda_LM::Optimize (include/slam/NonlinearSolver_Lambda_LM.h:796-1135) with
    the Levenberg trust-region policy of :151-222:
        alpha0 = 1e-3 * largest diagonal entry of any vertex Hessian;  last = chi2(x)
        loop: Lambda = J^T Omega J + alpha I (re-linearized only after an accepted step), dx = Lambda^-1 eta,
              stop if ||dx|| <= threshold; save x; x <- x (+) dx; err = chi2(x);
              rho = (last - err) / (dx . (alpha dx + eta));
              rho > 0: alpha *= max(1/3, 1 - (2 rho - 1)^3), nu = 2, last = err
              else   : alpha *= nu, nu *= 2, restore x, and the iteration budget grows by one (at most 10 times)
    `path`: _ResidentBAPath (default, GPU) or any object with the same methods (tests inject a host path)."""

    def __init__(self, system, path=None, device=0, verbose=False):
        self.system = system
        self.path = path if path is not None else _ResidentBAPath(device)
        self.verbose = verbose
        self.n_iterations = 0
        self.alpha = None
        self.chi2_history = []

    def Optimize(self, n_max_iteration_num=5, f_min_dx_norm=0.01):
        p = self.path
        p.begin(self.system)
        p.linearize()
        alpha = 1e-3 * p.max_hessian_diag()
        nu = 2.0
        last = p.chi2()
        self.chi2_history = [last]
        fail = 10
        dirty = False       # J / r on the device are those of the current state (chi2 re-linearized it)
        it = 0
        while it < n_max_iteration_num:
            if it and dirty:
                p.linearize()
            dirty = False
            ok, norm = p.solve(alpha)
            self.n_iterations = it + 1
            if not ok:
                break
            if self.verbose:
                print("iter %d: alpha %.6g ||dx|| %.6g" % (it, alpha, norm))
            if norm <= f_min_dx_norm:
                break
            p.save()
            denom = p.gain_denominator(alpha)
            p.apply()
            err = p.chi2()          # leaves J / r of the NEW state on the device
            rho = (last - err) / denom
            if rho > 0:
                alpha *= max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)
                nu = 2.0
                last = err
                self.chi2_history.append(err)
            else:
                alpha *= nu
                nu *= 2.0
                p.restore()
                dirty = True    # the device holds the linearization of the rejected state: redo it at the restored one
                if fail > 0:
                    fail -= 1
                    n_max_iteration_num += 1
            it += 1
        self.alpha = alpha
        p.finish(self.system)
        return self.n_iterations
