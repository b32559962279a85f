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
