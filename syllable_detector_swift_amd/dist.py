"""Multi-GPU sharding of a detector bank: channels are independent detectors
(Processor.swift:57-59, main.swift:86-89), so ranks own contiguous channel blocks and the
data path needs no collective.  The only exchange is ONE gather of the per-channel detection
flags per batch (RCCL over xGMI on GPUs; `gloo` in the CPU tests): a few MB, latency-bound,
so it is issued once per run, never per frame.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).
"""
from __future__ import annotations

from typing import Optional, Tuple


def shard_channels(total_channels: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of channels owned by `rank`: (first, count).  The first
    total % world ranks take one extra channel."""
    if total_channels < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad sharding arguments")
    base, extra = divmod(total_channels, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_flags(local_flags, total_channels: int, group=None):
    """All ranks receive the full [total_channels, E] flag tensor.  `local_flags` is this rank's
    [count, E] uint8 tensor (CUDA with nccl/RCCL, CPU with gloo).  Equal shards use one
    all_gather_into_tensor; ragged shards pad to the largest shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    E = int(local_flags.shape[1])
    counts = [shard_channels(total_channels, world, r)[1] for r in range(world)]
    assert local_flags.shape[0] == counts[rank], "local shard does not match shard_channels()"
    if len(set(counts)) == 1:
        out = torch.empty((total_channels, E), dtype=local_flags.dtype, device=local_flags.device)
        dist.all_gather_into_tensor(out, local_flags.contiguous(), group=group)
        return out
    biggest = max(counts)
    padded = torch.zeros((biggest, E), dtype=local_flags.dtype, device=local_flags.device)
    padded[: counts[rank]] = local_flags
    buf = torch.empty((world * biggest, E), dtype=local_flags.dtype, device=local_flags.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * biggest: r * biggest + counts[r]] for r in range(world)], dim=0)


class ShardedSyllableDetector:
    """This rank's share of a `total_channels`-wide bank on its own GPU."""

    def __init__(self, config, total_channels: int, device: Optional[int] = None, group=None, engine: int = 0):
        import torch.distributed as dist
        from .detector import SyllableDetector
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_channels = int(total_channels)
        self.first, self.count = shard_channels(self.total_channels, self.world, self.rank)
        if self.count == 0:
            raise ValueError("fewer channels than ranks: shard by time with a (T-1)*hop + W - hop halo instead")
        self.detector = SyllableDetector(config, channels=self.count, device=0 if device is None else device, engine=engine)

    def run(self, local_samples, gather: bool = True):
        """local_samples: this rank's [count, S] block.  Returns (outputs_local, flags) where flags is
        the gathered [total_channels, E] tensor when `gather`, else the local one."""
        outputs, flags = self.detector.run(local_samples)
        if gather and self.world > 1:
            flags = gather_flags(flags, self.total_channels, self.group)
        return outputs, flags

    def close(self):
        self.detector.close()
