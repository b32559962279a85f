"""GPU: the multi-GPU split API (spp_set_shard / spp_schur_form / all-reduce / spp_schur_finish) driven
on ONE device: two contexts own the two landmark shards, their partial S | rhs buffers are summed on
the host (standing in for the RCCL all-reduce of bench.py), each context finishes on the summed
buffer. The union of the shards' solutions must equal the unsharded solve."""
import numpy as np
import pytest

from slam_plus_plus_amd import api, synth
from oracle import spp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,world,mode", [("ba_small", 2, api.MODE_SCHUR), ("ba_interleaved", 3, api.MODE_SCHUR),
                                             ("ba_medium", 2, api.MODE_SCHUR), ("ba_small", 2, api.MODE_SCHUR_SPARSE),
                                             ("ba_banded", 3, api.MODE_SCHUR_SPARSE)])
def test_sharded_schur_equals_unsharded(name, world, mode):
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    full = api.CLinearSolver_HIP(mode=mode)
    xfull = eta.copy()
    assert full.Solve_PosDef_Blocky(lam, xfull)
    ctxs, bufs = [], []
    for r in range(world):
        c = api.Context(0)
        c.set_shard(r, world)
        c.analyze(lam, mode)
        assert c.info("MODE") == mode
        dv = api.DeviceArray.from_host(c, lam.vals)
        dr = api.DeviceArray.from_host(c, eta)
        dS = api.DeviceArray(c, c.schur_buffer_size())
        c.schur_form(dv.ptr, dr.ptr, dS.ptr)
        c.synchronize()
        ctxs.append(c)
        bufs.append((dv, dr, dS))
    assert sum(c.info("N_LANDMARKS") for c in ctxs) == int((lam.dim == 3).sum())
    # pack -> sum -> unpack: what bench.py sends through the RCCL all-reduce (upper trapezoid only)
    packed = []
    for c, (dv, dr, dS) in zip(ctxs, bufs):
        dP = api.DeviceArray(c, c.schur_packed_size())
        c.schur_pack(dS.ptr, dP.ptr)
        c.synchronize()
        packed.append(dP)
    if mode == api.MODE_SCHUR:
        nblk = ctxs[0].info("S_LD") // 128
        assert packed[0].n == 128 * 128 * nblk * (nblk + 1) // 2 <= bufs[0][2].n
    else:  # sparse reduced system: block values | rhs, the same structure (union over all landmarks) on every rank
        assert len({c.info("S_NNZB") for c in ctxs}) == 1 and ctxs[0].info("S_NNZB") == full.ctx.info("S_NNZB")
        assert packed[0].n == bufs[0][2].n == 36 * ctxs[0].info("S_NNZB") + ctxs[0].info("N_REDUCED")
    psum = sum(p.download() for p in packed)  # the all-reduce
    for c, (dv, dr, dS), dP in zip(ctxs, bufs, packed):
        dP.upload(psum)
        c.schur_unpack(dP.ptr, dS.ptr)
        c.synchronize()
    total = bufs[0][2].download()
    x = np.zeros_like(eta)
    dl = int(lam.dim.min())
    for r, (c, (dv, dr, dS)) in enumerate(zip(ctxs, bufs)):
        assert c.schur_finish(dv.ptr, dS.ptr, dr.ptr) == 0
        c.synchronize()
        xr = dr.download()
        if r == 0:
            for b in np.flatnonzero(lam.dim != dl):
                x[lam.base[b]:lam.base[b + 1]] = xr[lam.base[b]:lam.base[b + 1]]
        for b in orc.landmark_shard(lam, r, world):
            x[lam.base[b]:lam.base[b + 1]] = xr[lam.base[b]:lam.base[b + 1]]
        # every rank holds the same pose update
        pb = np.flatnonzero(lam.dim != dl)[0]
        assert np.allclose(xr[lam.base[pb]:lam.base[pb + 1]], xfull[lam.base[pb]:lam.base[pb + 1]], rtol=1e-9, atol=0)
    assert np.linalg.norm(x - xfull) / np.linalg.norm(xfull) < 1e-11
    for c, b in zip(ctxs, bufs):
        for d in b:
            d.free()
        c.close()
