"""Yardstick of the parity tests on ill-conditioned systems (test infrastructure).

north_star's bound is 1e-10 relative on Delta-x. The synthetic pose graphs are anchored by one unit prior only
(cond 1e7..1e11): there the reference's OWN backends are 1e-9..1e-7 away from the exact solution of the same
Lambda -- and their mutual spread understates that, because UberBlock, CSparse and CHOLMOD all eliminate in AMD
order and round alike. So the error of every solver is measured against a REFINED solution (sparse LU + iterative
refinement with the residual accumulated in extended precision: exact to ~cond * eps^2), and the test demands

    err(ours)  <=  max(1e-10, 4 * max over the reference backends of err(backend)),

i.e. our solver may be at most a small factor less accurate than the reference is itself. (The fixtures store
the backends' solutions, strided for the large systems: errors are taken over the stored entries.)"""
import numpy as np


def refined_solution(lam, eta, iters=4):
    import scipy.sparse.linalg as spla
    A = lam.to_scipy().tocsc()
    lu = spla.splu(A)
    C = A.tocoo()
    r_, c_, v_ = C.row, C.col, C.data.astype(np.longdouble)
    b = eta.astype(np.longdouble)
    x = lu.solve(eta).astype(np.longdouble)
    for _ in range(iters):
        res = b.copy()
        np.subtract.at(res, r_, v_ * x[c_])
        x = x + lu.solve(res.astype(np.float64)).astype(np.longdouble)
    return x.astype(np.float64)


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def reference_error(g, x_true):
    """largest error of the reference's backends stored in fixture g, over the stored (strided) entries"""
    s = int(g["dx_stride"])
    return max(rel(g[k], x_true[::s]) for k in g.files if k.startswith("dx_") and k != "dx_stride")


def check_against_reference(x, g, lam, eta, factor=4.0):
    x_true = refined_solution(lam, eta)
    s = int(g["dx_stride"])
    e_ref = reference_error(g, x_true)
    e = rel(x[::s], x_true[::s])
    assert e <= max(1e-10, factor * e_ref), (e, e_ref)
    return e, e_ref


def check_against_solutions(x, sols, lam, eta, factor=4.0):
    """the same criterion against live reference solutions {backend: full solution vector}"""
    x_true = refined_solution(lam, eta)
    e_ref = max(rel(v, x_true) for v in sols.values())
    e = rel(x, x_true)
    assert e <= max(1e-10, factor * e_ref), (e, e_ref)
    return e, e_ref
