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
from .formats import se2_linearize, se3_linearize, se3_plus


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
