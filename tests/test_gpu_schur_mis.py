"""GPU parity of the MIS Schur cut (SPP_MODE_SCHUR_MIS): on a graph of ONE block width a maximal independent set
of vertices is eliminated through the Schur complement (their diagonal part is block diagonal by construction),
the reduced system over the remaining vertices is kept sparse and solved by the supernodal path -- the general
ordering of the reference's CSchurOrdering (src/slam/LinearSolver_Schur.cpp:690-769,1235-1340) with a sparse inner
solver (include/slam/LinearSolver_Schur.h:1844-1853). The set is chosen greedily by degree (valid, deterministic,
not necessarily the reference's set: any independent set gives the same solution).

Tolerances as in tests/test_gpu_sparse.py: 1e-10 on well-conditioned (damped) systems; on the ill-conditioned
undamped pose graphs at most 4 x the reference backends' own error against the refined solution (tests/parity.py) plus a residual bound."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _refs(lam, eta):
    out = {}
    if orc.have_ref():
        for be in ("uberblock", "csparse", "cholmod"):
            st, x, _ = orc.RefSolver(be, lam).solve(lam.vals, eta)
            assert st == 0
            out[be] = x
    st, xo = orc.solve_blocky(lam, eta)
    assert st == 0
    out["oracle"] = xo
    return out


@pytest.mark.parametrize("name,damping", [("se2_small", 50.0), ("se3_small", 50.0), ("manhattan3500", 50.0), ("sphere2500", 50.0),
                                          ("se2_small", None), ("se3_small", None), ("manhattan3500", None), ("sphere2500", None)])
def test_mis_schur_matches_reference(name, damping):
    prob = synth.make(name)
    lam, eta = orc.assemble(prob) if damping is None else orc.assemble(prob, damping=damping)
    solver = api.CLinearSolver_HIP(mode=api.MODE_SCHUR_MIS)
    x = eta.copy()
    assert solver.Solve_PosDef_Blocky(lam, x)
    ctx = solver.ctx
    assert ctx.info("MODE") == api.MODE_SCHUR_MIS
    n_lm, n_red = ctx.info("N_LANDMARKS"), ctx.info("N_REDUCED")
    d = int(lam.dim[0])
    assert 0 < n_lm < lam.nb and n_red == d * (lam.nb - n_lm)
    # the eliminated set is independent: no stored off-diagonal block joins two of its members
    order = ctx.ordering(lam.nb)
    elim = set(order[lam.nb - n_lm:].tolist())
    cols = np.repeat(np.arange(lam.nb), np.diff(lam.col_ptr))
    off = lam.row_idx != cols
    assert not any((int(i) in elim) and (int(j) in elim) for i, j in zip(lam.row_idx[off], cols[off]))
    assert n_lm >= lam.nb // 4, "a maximal independent set of these sparse graphs is a sizeable share of the vertices"
    res = np.linalg.norm(lam.matvec(x) - eta) / np.linalg.norm(eta)
    assert res < 1e-11, res
    sols = _refs(lam, eta)
    if damping is not None:
        for k, xr in sols.items():
            assert _rel(x, xr) < 1e-10, (k, _rel(x, xr))
    else:
        import parity
        parity.check_against_solutions(x, sols, lam, eta)
    x2 = eta.copy()
    assert solver.Solve_PosDef_Blocky(lam, x2)
    assert np.array_equal(x, x2), "bit-reproducible"


def test_mis_mode_rejects_two_width_graphs_and_auto_never_picks_it():
    lam, eta = orc.assemble(synth.make("ba_small"))
    with pytest.raises(Exception):
        api.CLinearSolver_HIP(mode=api.MODE_SCHUR_MIS).Solve_PosDef_Blocky(lam, eta.copy())
    lam, eta = orc.assemble(synth.make("se2_small"))
    s = api.CLinearSolver_HIP(mode=api.MODE_AUTO)
    assert s.Solve_PosDef_Blocky(lam, eta.copy()) and s.ctx.info("MODE") == api.MODE_SPARSE
