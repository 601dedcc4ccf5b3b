"""Point-sharded MSM across the GPUs of one node (BASELINE.json config 5).

An MSM is a sum over (scalar, base) pairs, so rank r owns the contiguous shard [r*n/G, (r+1)*n/G): its bases stay
resident in its HBM, it runs the whole Pippenger locally and produces ONE partial result (a Jacobian point, 144 B).
The only exchange is an all-gather of G x 144 bytes (RCCL over xGMI when the backend is "nccl"; an EC-point sum is
not expressible as an RCCL reduction op, so "all-reduce" = all-gather + local group add).  Every rank then adds the
G partials in rank order, so all ranks hold identical bytes.  No other collective is on the data path."""
from __future__ import annotations
import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced shard of [0, n) for `rank` (first n % world ranks get one extra element)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_partials(partial: np.ndarray, group=None, device=None) -> np.ndarray:
    """partial: uint64[k] on the host (k = 18 for a Jacobian point) -> uint64[world,k] on every rank.
    With backend "nccl" (= RCCL) pass the rank's device: the 144-byte payload travels over xGMI."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])


def sharded_msm(local_msm, combine, partial_for_rank=None, group=None, device=None) -> np.ndarray:
    """local_msm() -> uint64[18] partial of this rank's shard; combine(uint64[G,18]) -> uint64[18].
    The product passes VariableBase.msm over the rank's pinned shard and aleo_amd.g1_sum."""
    part = local_msm()
    allp = all_gather_partials(part, group=group, device=device)
    return combine(allp)
