"""
Multi-GPU sharding of a batch of independent rollouts: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm), contiguous env-index ranges per rank, RNG counters keyed by the GLOBAL env index so that results do
not depend on the number of ranks, and one all-gather of the per-rollout results the algorithms consume.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int):
    """ [start, stop) of rank's contiguous share of n items; the first n % world_size ranks get one more. """
    q, r = divmod(n, world_size)
    start = rank*q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def all_gather_rows(t: torch.Tensor, n_total: int) -> torch.Tensor:
    """ Concatenate every rank's leading-dimension shard (sizes from shard_range(n_total, ...)) on all ranks.
    Shards of unequal size are padded to the largest one for the collective and trimmed afterwards. """
    rank, ws = world()
    if ws == 1:
        return t
    sizes = [shard_range(n_total, r, ws)[1] - shard_range(n_total, r, ws)[0] for r in range(ws)]
    assert t.shape[0] == sizes[rank], (t.shape, sizes, rank)
    m = max(sizes)
    t = t.contiguous()
    if t.shape[0] < m:
        pad = torch.zeros((m - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad])
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    if len(set(sizes)) == 1:
        return torch.cat(parts)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])
