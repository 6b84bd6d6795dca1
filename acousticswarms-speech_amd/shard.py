"""Candidate sharding across the GPUs of one node.

Candidates of one stage share only the read-only mixture and the weights, so they are
partitioned with no data-path collective; the single exchange step of a stage is an
all-gather of the per-candidate energies ([N,2] float64, a few KB) before the host-side
thresholding / clustering.  One process per GPU, ``torch.distributed`` (backend "nccl"
is RCCL over xGMI; "gloo" on CPU for tests).  This replaces the per-call replicate /
scatter / gather of ``nn.DataParallel`` (sep/training/JointModel/network.py:30,93):
weights are resident per rank, only offsets and energies move.
"""
from typing import Callable, List, Sequence

import numpy as np


def shard_bounds(n_items: int, world: int) -> List[int]:
    """Balanced contiguous partition: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n_items, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


def shard_groups(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Partition whole groups (all fine candidates of one surviving coarse patch stay on
    one rank so its SI-SDR clustering is local, sep/Mic_Array.py:339-383) with a
    longest-processing-time greedy balance.  Returns group indices per rank."""
    order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
    load = [0] * world
    owner = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += sizes[i]
    return [sorted(o) for o in owner]


class ShardedScorer:
    """score(mix, offsets[N,P]) -> energies [N,2] on every rank.

    ``local_score(mix, offsets_local) -> ndarray|tensor [n_local,2] float64`` is the
    single-GPU scorer (SpotModel.shift_and_score bound to a window)."""

    def __init__(self, local_score: Callable, group=None):
        import torch.distributed as dist
        self.local_score = local_score
        self.group = group
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1

    def score(self, mix, offsets: np.ndarray, device=None) -> np.ndarray:
        import torch
        N = int(offsets.shape[0])
        b = shard_bounds(N, self.world)
        lo, hi = b[self.rank], b[self.rank + 1]
        local = self.local_score(mix, offsets[lo:hi])
        local = torch.as_tensor(local, dtype=torch.float64)
        if self.world == 1:
            return local.cpu().numpy().reshape(N, 2)
        width = max(b[r + 1] - b[r] for r in range(self.world))     # pad to equal shards
        dev = device if device is not None else local.device
        buf = torch.zeros((width, 2), dtype=torch.float64, device=dev)
        buf[:hi - lo] = local.to(dev)
        out = torch.empty((self.world * width, 2), dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)   # the stage's one exchange
        out = out.view(self.world, width, 2).cpu().numpy()
        return np.concatenate([out[r, :b[r + 1] - b[r]] for r in range(self.world)], axis=0)
