"""The one data-path exchange of the landmark-sharded Schur solve: every rank holds a packed partial reduced camera
system (+ reduced right-hand side); afterwards every rank holds their sum (the reference has no multi-GPU
path; CLinearSolver_Schur forms S on one device, include/slam/LinearSolver_Schur.h:1767-1843).

xGMI on an MI355X node is point to point (7 links per GPU, ~153 GB/s each): a ring all-reduce of B bytes moves
2 (N - 1) / N x B over ONE link per GPU. `direct` uses all links at once:
    reduce-scatter as ONE all-to-all   rank r sends slice j of its buffer to rank j (B / N per link),
    local sum of the N received slices in RANK ORDER (deterministic: the same bits on every run),
    all-gather of the summed slices    (B / N per link again).
`torch.distributed` only (backend "nccl" = RCCL on ROCm; gloo for CPU rehearsals): no torch types cross the C ABI."""
import torch
import torch.distributed as dist


class PackedExchange:
    def __init__(self, numel, world, device, mode="direct"):
        self.world = world
        self.mode = mode if world > 1 else "none"
        self.notes = []
        self.padded = (numel + world - 1) // world * world  # equal slices; the tail stays zero
        self.buf = torch.zeros(self.padded, dtype=torch.float64, device=device)
        if self.mode == "direct":
            self.recv = torch.empty(self.padded, dtype=torch.float64, device=device)
            self.own = torch.empty(self.padded // world, dtype=torch.float64, device=device)

    def sum(self):
        """in place on self.buf; ordered against the current stream like any torch collective"""
        if self.mode == "none":
            return
        if self.mode == "direct":
            try:
                dist.all_to_all_single(self.recv, self.buf)
                torch.sum(self.recv.view(self.world, -1), dim=0, out=self.own)
                dist.all_gather_into_tensor(self.buf, self.own)
                return
            except Exception as e:  # noqa: BLE001 -- a backend without all-to-all: the ring all-reduce instead
                self.mode = "allreduce"
                self.notes.append("direct exchange failed (%r): all_reduce" % (e,))
        dist.all_reduce(self.buf)
