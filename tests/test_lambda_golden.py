"""The ASSEMBLY restatement pinned at the level of Lambda and eta themselves: tests/golden/{se2,se3,ba}_lambda.npz hold,
for three small graphs, the per-edge J0 / J1 / information / error the reference's edge types computed AND the Lambda
(upper block triangle) and eta the reference's nonlinear solver handed to its linear solver after Refresh_Lambda
(include/slam/NonlinearSolver_Lambda_Base.h:1658-1688, include/slam/BaseTypes_Binary.h:759-848) -- recorded by
oracle/lambda_dump.cpp, converted by tools/make_golden_lambda.py. A transposed block, a wrong reduction order or a
misplaced unary factor shows here directly, not through a converged Gauss-Newton state.

CPU part: oracle/spp_oracle.c (orc_edge_hessians / orc_reduce) against the fixtures, <= 1e-13 of the largest entry.
GPU part (-m gpu): the HIP assembly kernels (csrc/spp_assemble.hip) against the same fixtures through the C ABI.
The BA graph interleaves camera and point ids: about half of its camera-point blocks are stored transposed
(BaseTypes_Binary.h:783-806); its second record carries the Levenberg-Marquardt damping on the diagonal
(NonlinearSolver_Lambda_LM.h:228-239).
ba_robust_lambda.npz: the same BA problem with ROBUST edges -- the reference's CBaseEdge::Robust option with its Huber
kernel (include/slam/RobustUtils.h), per-edge weights in `w` -- and the Lambda / eta of the reference's robust assembly
branch (BaseTypes_Binary.h:768-848: the weight once on every Hessian block, TWICE on the first vertex's right-hand side)."""
import os

import numpy as np
import pytest

from slam_plus_plus_amd import synth
from oracle import spp_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-13


def _load(kind):
    g = np.load(os.path.join(GOLD, kind + "_lambda.npz"))
    prob = synth.Problem(name=kind + "_lambda", dim=g["dim"].astype(np.int32), v0=g["v0"], v1=g["v1"], d0=int(g["d0"]),
                         d1=int(g["d1"]), rd=int(g["rd"]), J0=g["J0"], J1=g["J1"], Om=g["Om"], r=g["r"],
                         unary_vertex=0,  # __AUTO_UNARY_FACTOR_ON_VERTEX_ZERO (FlatSystem.h:331-337): here a LANDMARK for ba
                         damping=0.0)
    return g, prob


def _check(lam, eta, g, sfx=""):
    assert np.array_equal(lam.col_ptr, g["col_ptr"])
    assert np.array_equal(lam.row_idx, g["row_idx"])
    assert np.array_equal(lam.blk_off, g["blk_off"])
    vals, eta_ref = g["vals" + sfx], g["eta" + sfx]
    assert np.abs(lam.vals - vals).max() <= TOL * np.abs(vals).max(), np.abs(lam.vals - vals).max() / np.abs(vals).max()
    assert np.abs(eta - eta_ref).max() <= TOL * np.abs(eta_ref).max(), np.abs(eta - eta_ref).max() / np.abs(eta_ref).max()


def _lm_alpha(g):
    """the damping the reference added: Lambda_LM - Lambda is alpha on every diagonal entry and zero elsewhere"""
    diff = g["vals_lm"] - g["vals"]
    dim, col_ptr, blk_off = g["dim"], g["col_ptr"], g["blk_off"]
    mask = np.zeros(diff.size, dtype=bool)
    for j in range(dim.size):
        p = col_ptr[j + 1] - 1  # the diagonal block is the last block of its column
        d = int(dim[j])
        mask[blk_off[p] + np.arange(d) * (d + 1)] = True
    alpha = float(np.median(diff[mask]))
    assert alpha > 0
    assert np.abs(diff[mask] - alpha).max() <= 1e-12 * alpha and np.abs(diff[~mask]).max() == 0.0
    return alpha


@pytest.mark.parametrize("kind", ["se2", "se3", "ba"])
def test_oracle_assembly_matches_the_reference_lambda(kind):
    g, prob = _load(kind)
    lam, eta = orc.assemble(prob)
    _check(lam, eta, g)


def test_fixture_has_transposed_blocks_and_full_information():
    """the cases the fixture exists for: reversed vertex order (transposed off-diagonal block) and a non-diagonal Omega"""
    g, prob = _load("ba")
    rev = np.count_nonzero(g["v0"] > g["v1"])
    assert 0 < rev < g["v0"].size, rev
    Om = g["Om"].reshape(-1, 2, 2)
    assert np.all(Om[:, 0, 1] != 0)


def test_oracle_damped_assembly_matches_the_reference_lm_lambda():
    g, prob = _load("ba")
    alpha = _lm_alpha(g)
    lam, eta = orc.assemble(prob, damping=alpha)
    _check(lam, eta, g, "_lm")
    assert np.array_equal(g["eta_lm"], g["eta"])  # same state: damping does not touch eta


def test_oracle_robust_assembly_matches_the_reference_lambda():
    g, prob = _load("ba_robust")
    w = g["w"]
    assert np.count_nonzero(w < 1.0) > w.size // 3 and w.min() > 0 and w.max() == 1.0  # the kernel really bites
    lam, eta = orc.assemble(prob, weights=w)
    _check(lam, eta, g)
    lam0, eta0 = orc.assemble(prob)  # ... and the plain assembly is NOT what the fixture holds
    assert np.abs(lam0.vals - g["vals"]).max() > 1e-3 * np.abs(g["vals"]).max()


@pytest.mark.gpu
def test_hip_robust_assembly_matches_the_reference_lambda(hip_ctx):
    from slam_plus_plus_amd import api
    g, prob = _load("ba_robust")
    st = hip_ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    arrs = [api.DeviceArray.from_host(hip_ctx, np.ascontiguousarray(a, dtype=np.float64).ravel())
            for a in (prob.J0, prob.J1, prob.Om, prob.r, g["w"])]
    dv = api.DeviceArray(hip_ctx, st.nvals)
    de = api.DeviceArray(hip_ctx, st.n)
    hip_ctx.assemble_set_edge_weights(arrs[4].ptr)
    hip_ctx.assemble_device(*[a.ptr for a in arrs[:4]], 0.0, dv.ptr, de.ptr)
    _check(st.with_vals(dv.download()), de.download(), g)
    # the weights themselves from the device-resident errors: the fixture's edges use a Huber kernel at a scale of 3/4
    dw = api.DeviceArray(hip_ctx, g["w"].size)
    hip_ctx.edge_robust_weights_device(g["w"].size, prob.rd, arrs[3].ptr, dw.ptr, scale=0.75)
    w_dev = dw.download()
    dw.free()
    assert np.abs(w_dev - g["w"]).max() <= 4e-16, np.abs(w_dev - g["w"]).max()
    hip_ctx.assemble_set_edge_weights(None)  # back to plain edges: the plain fixture of the same problem
    hip_ctx.assemble_device(*[a.ptr for a in arrs[:4]], 0.0, dv.ptr, de.ptr)
    g0, _ = _load("ba")
    _check(st.with_vals(dv.download()), de.download(), g0)
    for d in arrs + [dv, de]:
        d.free()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["se2", "se3", "ba", "ba_lm"])
def test_hip_assembly_matches_the_reference_lambda(hip_ctx, kind):
    from slam_plus_plus_amd import api
    g, prob = _load(kind.split("_")[0])
    damping = _lm_alpha(g) if kind.endswith("_lm") else 0.0
    st = hip_ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    arrs = [api.DeviceArray.from_host(hip_ctx, np.ascontiguousarray(a, dtype=np.float64).ravel())
            for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    dv = api.DeviceArray(hip_ctx, st.nvals)
    de = api.DeviceArray(hip_ctx, st.n)
    hip_ctx.assemble_device(*[a.ptr for a in arrs], damping, dv.ptr, de.ptr)
    lam, eta = st.with_vals(dv.download()), de.download()
    for d in arrs + [dv, de]:
        d.free()
    _check(lam, eta, g, "_lm" if kind.endswith("_lm") else "")
