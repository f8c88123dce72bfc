"""Data-parallel sharding of the hot path.

Sequences of a batch are independent (every kernel indexes by batch element; eval-mode BatchNorm
uses running statistics), so a global batch splits contiguously over ranks -- one process per GPU --
with NO collective on the data path.  The reference's only multi-GPU mechanism is nn.DataParallel in
training (train.py:73-80: per-step weight broadcast + scatter + gather inside one process); here the
weights are generated/loaded locally on every rank and the single exchange is the final all_gather
of the interpolated frames (2.36 MB per rank at N=8192, B=8) over RCCL/xGMI, or gloo on CPU.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    """Contiguous [lo, hi) of the global batch owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, rank, world):
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi].contiguous()


def pack_frames(out_lst):
    """3 x (B,N,3) -> (B,3,N,3): one contiguous payload for the collective."""
    return torch.stack(out_lst, dim=1).contiguous()


def gather_frames(out_lst, world=None):
    """Final gather of the interpolated frames: every rank returns (B_global,3,N,3).
    Equal per-rank batch sizes (the bench's weak-scaling layout)."""
    local = pack_frames(out_lst)
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return local
    full = torch.empty((world * local.shape[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local)
    return full


def gather_frames_uneven(out_lst, global_batch):
    """Same for ragged shards (global batch not divisible by the world size)."""
    local = pack_frames(out_lst)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = []
    for r in range(world):
        lo, hi = shard_range(global_batch, r, world)
        buf = local if r == rank else torch.empty((hi - lo, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, dim=0)
