"""CPU, world_size 2, gloo: the multi-GPU decomposition of the Schur path.
Landmarks are sharded round-robin over the ranks; every rank forms its partial reduced camera system
(rank 0 carries A and the pose rhs), ONE all-reduce sums S | rhs, every rank solves the reduced system
redundantly and back-substitutes its own landmarks. The HIP kernels cannot run here, so the per-rank
partials come from the CPU oracle; what is tested is the sharding rule, the collective and the
assembly of the distributed solution -- the same sequence bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slam_plus_plus_amd import synth
from oracle import spp_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = synth.make(name)
    lam, eta = orc.assemble(prob)
    S, xred, pose_idx, mine = orc.schur_partial(lam, eta, rank, world)
    n_p = S.shape[0]
    buf = torch.from_numpy(np.concatenate([np.triu(S).ravel(order="F"), xred]))
    dist.all_reduce(buf)  # the single data-path collective
    S_sum = buf[:n_p * n_p].numpy().reshape((n_p, n_p), order="F")
    x_sum = buf[n_p * n_p:].numpy()
    Sd = np.triu(S_sum) + np.triu(S_sum, 1).T
    dx = np.linalg.solve(Sd, x_sum)  # every rank factors S redundantly
    mine_dx = orc.schur_backsubstitute(lam, eta, dx, mine)
    # gather the landmark pieces on rank 0 for the check only (the product keeps them sharded)
    pieces = [None] * world
    dist.all_gather_object(pieces, {int(b): v for b, v in mine_dx.items()})
    if rank == 0:
        x = np.zeros_like(eta)
        x[pose_idx] = dx
        for d in pieces:
            for b, v in d.items():
                x[lam.base[b]:lam.base[b] + v.size] = v
        st, xfull, _ = orc.schur_solve(lam, eta)
        ret["rel"] = float(np.linalg.norm(x - xfull) / np.linalg.norm(xfull))
        ret["shards"] = [len(d) for d in pieces]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ba_small", "ba_interleaved"])
def test_two_rank_landmark_sharding_reproduces_the_full_solve(name):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), name, ret), nprocs=world, join=True)
    assert ret["rel"] < 1e-10, ret["rel"]
    assert abs(ret["shards"][0] - ret["shards"][1]) <= 1  # round-robin balance


def _xch_worker(rank, world, port, numel, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slam_plus_plus_amd.exchange import PackedExchange
    rng = np.random.default_rng(100 + rank)
    mine = rng.standard_normal(numel)
    out = {}
    for mode in ("direct", "allreduce"):
        x = PackedExchange(numel, world, "cpu", mode)
        assert x.padded % world == 0 and x.padded >= numel
        for _ in range(2):  # twice: the buffers are reused call after call
            x.buf.zero_()
            x.buf[:numel] = torch.from_numpy(mine)
            x.sum()
        out[mode] = (x.buf.numpy().copy(), x.mode, list(x.notes))
    want = sum(np.random.default_rng(100 + r).standard_normal(numel) for r in range(world))
    gathered = [None] * world
    dist.all_gather_object(gathered, out["direct"][0][:numel].tobytes())
    if rank == 0:
        ret["direct_mode"] = out["direct"][1]
        ret["err_direct"] = float(np.abs(out["direct"][0][:numel] - want).max())
        ret["err_allreduce"] = float(np.abs(out["allreduce"][0][:numel] - want).max())
        ret["tail_zero"] = bool(np.all(out["direct"][0][numel:] == 0))
        ret["same_on_all_ranks"] = all(g == gathered[0] for g in gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,numel", [(2, 1001), (3, 4096)])
def test_direct_exchange_equals_the_all_reduce(world, numel):
    """slam_plus_plus_amd/exchange.py (what bench.py --gpus N runs over RCCL): reduce-scatter as one all-to-all + local
    sum in rank order + all-gather gives the all-reduce's sum, identical bits on every rank, also when the buffer does
    not divide by the number of ranks (zero padding)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_xch_worker, args=(world, _free_port(), numel, ret), nprocs=world, join=True)
    assert ret["direct_mode"] == "direct", "gloo offers all_to_all_single on CPU tensors: the direct path must have run"
    assert ret["err_direct"] < 1e-13 and ret["err_allreduce"] < 1e-13
    assert ret["tail_zero"] and ret["same_on_all_ranks"]
