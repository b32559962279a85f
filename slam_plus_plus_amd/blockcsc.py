"""Flat upper-triangular block-CSC: the flattening of the reference's CUberBlockMatrix
(reference include/slam/BlockMatrixBase.h:380-503, element order BlockMatrix.cpp:3764-3807) that
crosses the C ABI (include/spp_hip.h). Pure numpy; used by tests, bench and the Python mirror of the
solver concept.
"""
import numpy as np


class BlockCSC:
    def __init__(self, dim, col_ptr, row_idx, blk_off, vals, nvals=None):
        """nvals: the length of the value array when the caller knows it (the library reports it); col_idx (the column of
        every block) and nvals are otherwise derived on first use -- on a Venice-sized structure (3.4 M blocks) those
        passes cost more than the library's whole assembly plan"""
        self.dim = np.ascontiguousarray(dim, dtype=np.int32)
        self.col_ptr = np.ascontiguousarray(col_ptr, dtype=np.int64)
        self.row_idx = np.ascontiguousarray(row_idx, dtype=np.int64)
        self.blk_off = np.ascontiguousarray(blk_off, dtype=np.int64)
        self.vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.float64)
        self.nb = int(self.dim.size)
        self.nnzb = int(self.row_idx.size)
        self.base = np.zeros(self.nb + 1, dtype=np.int64)
        np.cumsum(self.dim, out=self.base[1:])
        self.n = int(self.base[-1])
        self._col_idx = None
        self._nvals = None if nvals is None else int(nvals)

    @property
    def col_idx(self):
        if self._col_idx is None:
            self._col_idx = np.repeat(np.arange(self.nb, dtype=np.int64), np.diff(self.col_ptr))
        return self._col_idx

    @property
    def nvals(self):
        if self._nvals is None:
            self._nvals = int((self.blk_off + self.dim[self.row_idx].astype(np.int64) * self.dim[self.col_idx]).max()) if self.nnzb else 0
        return self._nvals

    def with_vals(self, vals):
        return BlockCSC(self.dim, self.col_ptr, self.row_idx, self.blk_off, vals, self._nvals)

    def to_dense(self, symmetric=True):
        """Dense n x n matrix (mirrors the upper triangle when symmetric=True). Small cases only."""
        A = np.zeros((self.n, self.n))
        for p in range(self.nnzb):
            i, j = int(self.row_idx[p]), int(self.col_idx[p])
            di, dj = int(self.dim[i]), int(self.dim[j])
            blk = self.vals[self.blk_off[p]:self.blk_off[p] + di * dj].reshape(dj, di).T
            A[self.base[i]:self.base[i] + di, self.base[j]:self.base[j] + dj] = blk
            if symmetric and i != j:
                A[self.base[j]:self.base[j] + dj, self.base[i]:self.base[i] + di] = blk.T
        if symmetric:
            # diagonal blocks: the reference stores them fully symmetric; use the upper half
            A = np.triu(A) + np.triu(A, 1).T
        return A

    def to_scipy(self):
        """scipy.sparse CSC of the full symmetric matrix (for residual checks at any size)."""
        import scipy.sparse as sp
        di = self.dim[self.row_idx].astype(np.int64)
        dj = self.dim[self.col_idx].astype(np.int64)
        cnt = di * dj
        tot = int(cnt.sum())
        blk = np.repeat(np.arange(self.nnzb), cnt)
        start = np.zeros(self.nnzb + 1, dtype=np.int64)
        np.cumsum(cnt, out=start[1:])
        e = np.arange(tot, dtype=np.int64) - start[blk]
        r = self.base[self.row_idx][blk] + e % di[blk]
        c = self.base[self.col_idx][blk] + e // di[blk]
        v = self.vals[self.blk_off[blk] + e]
        keep = r <= c  # upper triangle of the diagonal blocks only
        r, c, v = r[keep], c[keep], v[keep]
        U = sp.csc_matrix((v, (r, c)), shape=(self.n, self.n))
        return U + sp.triu(U, 1).T

    def matvec(self, x):
        return self.to_scipy() @ x


def structure_from_pairs(dim, rows, cols):
    """Upper block pattern from (row <= col) block coordinates (duplicates allowed) plus every
    diagonal block. Returns (BlockCSC without values, index of each input pair in the block list).
    Blocks are laid out contiguously in column order: blk_off = cumsum of block sizes."""
    dim = np.ascontiguousarray(dim, dtype=np.int32)
    nb = dim.size
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    assert np.all(rows <= cols)
    allr = np.concatenate([rows, np.arange(nb, dtype=np.int64)])
    allc = np.concatenate([cols, np.arange(nb, dtype=np.int64)])
    key = allc * nb + allr
    uniq, inv = np.unique(key, return_inverse=True)
    urow, ucol = uniq % nb, uniq // nb
    col_ptr = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(col_ptr, ucol + 1, 1)
    np.cumsum(col_ptr, out=col_ptr)
    size = dim[urow].astype(np.int64) * dim[ucol]
    blk_off = np.zeros(uniq.size, dtype=np.int64)
    np.cumsum(size[:-1], out=blk_off[1:])
    return BlockCSC(dim, col_ptr, urow, blk_off, None), inv[:rows.size], inv[rows.size:]
