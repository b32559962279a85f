"""Host-side fill-reducing orderings (spp_block_ordering; no GPU): validity, fill against the
reference's AMD (amd_l2 through CMatrixOrdering::p_BlockOrdering, src/slam/OrderingMagic.cpp:701 --
compiled into oracle/_ref) and the tree-height property nested dissection exists for.
Fill is counted by the oracle's symbolic up-looking factorization (BlockMatrix.cpp:9403-9545)."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from slam_plus_plus_amd.blockcsc import structure_from_pairs
from oracle import spp_oracle as orc


def _fill_and_height(lam, order):
    perm = orc.permute_upper(lam, order)
    parent = orc.etree(perm)
    r_col_ptr = np.empty(perm.nb + 1, dtype=np.int64)
    from oracle.spp_oracle import lib, i64, _p
    nnzb = lib().orc_chol_symbolic(i64(perm.nb), _p(perm.col_ptr), _p(perm.row_idx), _p(parent), _p(r_col_ptr), None)
    depth = np.ones(perm.nb, dtype=np.int64)
    for j in range(perm.nb):
        p = parent[j]
        if p >= 0:
            depth[p] = max(depth[p], depth[j] + 1)
    return int(nnzb), int(depth.max())


def _ring_band(nb, half):
    j = np.repeat(np.arange(nb), half)
    i = (j + np.tile(np.arange(1, half + 1), nb)) % nb
    dim = np.full(nb, 3, dtype=np.int32)
    st, _, _ = structure_from_pairs(dim, np.concatenate([np.minimum(i, j), np.arange(nb)]),
                                    np.concatenate([np.maximum(i, j), np.arange(nb)]))
    return st.with_vals(np.zeros(st.nvals))


@pytest.mark.parametrize("name", ["se2_small", "se3_small", "manhattan3500", "sphere2500", "ba_small"])
def test_orderings_are_permutations_with_fill_close_to_the_reference_amd(name):
    prob = synth.make(name)
    lam, _ = orc.assemble(prob)
    natural, _ = _fill_and_height(lam, np.arange(lam.nb))
    fills = {}
    for method in (api.ORDER_AMD, api.ORDER_ND):
        order = api.block_ordering(lam, method)
        assert sorted(order.tolist()) == list(range(lam.nb))
        fills[method], _ = _fill_and_height(lam, order)
    assert fills[api.ORDER_AMD] <= natural
    if orc.have_ref():
        rs = orc.RefSolver("uberblock", lam)
        inv = rs.ordering()            # the reference hands out the inverse ordering (destination of column i)
        ref_order = np.argsort(inv)
        ref_fill, _ = _fill_and_height(lam, ref_order)
        rs.close()
        assert fills[api.ORDER_AMD] <= 1.25 * ref_fill, (fills, ref_fill)
        assert fills[api.ORDER_ND] <= 2.5 * ref_fill, (fills, ref_fill)


def test_nested_dissection_flattens_the_tree_of_a_chain_like_graph():
    lam = _ring_band(3000, 8)
    amd = api.block_ordering(lam, api.ORDER_AMD)
    nd = api.block_ordering(lam, api.ORDER_ND)
    f_amd, h_amd = _fill_and_height(lam, amd)
    f_nd, h_nd = _fill_and_height(lam, nd)
    assert h_amd > 1000          # minimum degree: one long chain
    assert h_nd < h_amd / 10     # nested dissection: logarithmic in the number of subdomains
    assert f_nd < 1.3 * f_amd


def test_block_ordering_rejects_bad_input():
    lam = _ring_band(10, 2)
    lib = api.load_library()
    out = np.empty(10, dtype=np.int64)
    assert lib.spp_block_ordering(0, api._ptr(lam.col_ptr), api._ptr(lam.row_idx), 0, api._ptr(out)) < 0
    assert lib.spp_block_ordering(10, api._ptr(lam.col_ptr), api._ptr(lam.row_idx), 7, api._ptr(out)) < 0
    bad = lam.row_idx.copy()
    bad[0] = 9   # below the diagonal of column 0
    assert lib.spp_block_ordering(10, api._ptr(lam.col_ptr), api._ptr(bad), 0, api._ptr(out)) < 0
