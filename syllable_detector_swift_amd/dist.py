"""Multi-GPU sharding of a detector bank: channels are independent detectors
(Processor.swift:57-59, main.swift:86-89), so ranks own contiguous channel blocks and the
data path needs no collective.  The only exchange is ONE gather of the per-channel detection
flags per batch (RCCL over xGMI on GPUs; `gloo` in the CPU tests): a few MB, latency-bound,
so it is issued once per run, never per frame.

With fewer channels than ranks the TIME axis is sharded instead: a channel's evaluations are cut into contiguous
ranges, one per rank of that channel, and each rank reads its range's samples plus a halo of (T - 1) hop + W - hop samples --
the frames its last evaluations' windows reach into (SyllableDetector.swift:153-217: evaluation e is frames e .. e + T - 1,
frame j is samples [j hop + gap, j hop + gap + W)).  Still no collective on the data path; the one gather concatenates along time.

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


def shard_plane(total_channels: int, world_size: int, rank: int) -> Tuple[int, int, int, int]:
    """(first channel, channel count, part, parts) of `rank`.  With at least as many channels as ranks this is
    shard_channels() with one part; with fewer, every channel is shared by world // channels ranks (the first
    world % channels channels by one more) and `part` of `parts` is this rank's place along the channel's time axis."""
    if total_channels < 1 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad sharding arguments")
    if total_channels >= world_size:
        first, count = shard_channels(total_channels, world_size, rank)
        return first, count, 0, 1
    base, extra = divmod(world_size, total_channels)
    r = rank
    for ch in range(total_channels):
        parts = base + (1 if ch < extra else 0)
        if r < parts:
            return ch, 1, r, parts
        r -= parts
    raise AssertionError("unreachable")


def shard_evaluations(evaluations: int, parts: int, part: int) -> Tuple[int, int]:
    """Contiguous range of a channel's evaluations owned by `part`: (first, count); the first evaluations % parts
    parts take one more."""
    if evaluations < 0 or parts < 1 or not (0 <= part < parts):
        raise ValueError("bad sharding arguments")
    base, extra = divmod(evaluations, parts)
    return part * base + min(part, extra), base + (1 if part < extra else 0)


def time_shard_samples(hop: int, gap: int, window: int, time_range: int, first_eval: int, count: int) -> Tuple[int, int]:
    """Samples [s0, s1) a rank reads for evaluations [first_eval, first_eval + count): from its first frame's hop to the end
    of its last evaluation's last frame.  Neighbouring ranges overlap by the halo (time_range - 1) hop + window - hop
    (+ gap); a run over the slice yields exactly `count` evaluations, the same numbers the unsharded run gives them."""
    if count <= 0:
        return first_eval * hop, first_eval * hop
    return first_eval * hop, (first_eval + count + time_range - 2) * hop + gap + window


def gather_time_shards(local_flags, total_channels: int, evaluations: int, group=None):
    """The time-sharded counterpart of gather_flags: this rank holds [1, n] flags of its range of its channel; every rank
    receives [total_channels, evaluations].  Ranges are ragged by at most one evaluation: one padded all-gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    plan = [shard_plane(total_channels, world, r) for r in range(world)]
    spans = [shard_evaluations(evaluations, parts, part) for (_, _, part, parts) in plan]
    assert tuple(local_flags.shape) == (1, spans[rank][1]), "local flags do not match shard_plane() / shard_evaluations()"
    biggest = max(n for _, n in spans)
    padded = torch.zeros((biggest,), dtype=local_flags.dtype, device=local_flags.device)
    padded[: spans[rank][1]] = local_flags[0]
    buf = torch.empty((world * biggest,), dtype=local_flags.dtype, device=local_flags.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    out = torch.empty((total_channels, evaluations), dtype=local_flags.dtype, device=local_flags.device)
    for r, ((ch, _, _, _), (e0, n)) in enumerate(zip(plan, spans)):
        out[ch, e0: e0 + n] = buf[r * biggest: r * biggest + n]
    return out


def pack_flags(flags, out=None):
    """[rows, E] uint8 flags on a GPU -> [rows, ceil(E / 8)] bytes, bit b of byte t = flag 8 t + b (libsyldet kernel,
    on the current stream; `out`: a buffer to write into)."""
    import torch
    from . import _abi
    from .config import check
    rows, E = int(flags.shape[0]), int(flags.shape[1])
    flags = flags.contiguous()
    bits = out if out is not None else torch.empty((rows, (E + 7) // 8), dtype=torch.uint8, device=flags.device)
    assert bits.shape == (rows, (E + 7) // 8) and bits.is_contiguous() and bits.dtype == torch.uint8
    check(_abi.lib.syldet_pack_flags_device(flags.data_ptr(), rows, E, bits.data_ptr(),
                                            torch.cuda.current_stream(flags.device).cuda_stream))
    return bits


def unpack_flags(bits, E: int, out=None):
    """Inverse of pack_flags: [rows, ceil(E / 8)] bytes -> [rows, E] uint8 flags (0 / 1)."""
    import torch
    from . import _abi
    from .config import check
    rows = int(bits.shape[0])
    bits = bits.contiguous()
    flags = out if out is not None else torch.empty((rows, E), dtype=torch.uint8, device=bits.device)
    assert flags.shape == (rows, E) and flags.is_contiguous() and flags.dtype == torch.uint8
    check(_abi.lib.syldet_unpack_flags_device(bits.data_ptr(), rows, E, flags.data_ptr(),
                                              torch.cuda.current_stream(bits.device).cuda_stream))
    return flags


def gather_flags(local_flags, total_channels: int, group=None, packed=None):
    """All ranks receive the full [total_channels, E] flag tensor.  `local_flags` is this rank's
    [count, E] uint8 tensor (CUDA with nccl/RCCL, CPU with gloo).  Equal shards use one
    all_gather_into_tensor; ragged shards pad to the largest shard.  On GPUs the flags travel as
    bits (`packed`, default for CUDA tensors): an eighth of the bytes on the xGMI links, packed and
    unpacked by two small kernels on either side of the one collective."""
    import torch
    import torch.distributed as dist
    if packed is None:
        packed = bool(local_flags.is_cuda)
    if packed:
        E = int(local_flags.shape[1])
        bits = gather_flags(pack_flags(local_flags), total_channels, group, packed=False)
        return unpack_flags(bits, E)
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


class PipelinedFlagGather:
    """gather_flags for a stream of batches on GPUs with equal shards.  The whole exchange of batch i -- bit-packing, the one
    all-gather, the unpacking -- runs on a side stream under the kernels of batch i+1 instead of between them: xGMI is
    point-to-point, the ring all-gather of 8 ranks is per-link bound, and it is the only thing a rank ever waits for.  (Round 5:
    the packing too.  As a kernel of its own between two batches' kernels it cost the compute stream ~19 us a batch against ~3 us
    between back-to-back kernels.)  The packing READS the batch's flags after the compute stream has moved on, so a kernel that
    writes the same tensor again must wait for it: call `before_run(flags)` in front of that kernel -- with two flags tensors
    taken in turn the wait is for the packing of two batches ago, i.e. none.  Buffers alternate between two sets; `result(k)`
    makes the current stream wait for exchange k."""

    def __init__(self, local_rows: int, E: int, total_channels: int, device, group=None):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(group)
        if local_rows * world != total_channels:
            raise ValueError("PipelinedFlagGather needs equal shards (use gather_flags for ragged ones)")
        self.E, self.group, self.device = int(E), group, torch.device(device)
        nb = (self.E + 7) // 8
        self.bits = [torch.empty((local_rows, nb), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.gbits = [torch.empty((total_channels, nb), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.out = [torch.empty((total_channels, self.E), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.side = torch.cuda.Stream(self.device)
        self.done = [None, None]
        self.packed = [None, None]           # (event, data_ptr of the flags tensor the packing read)
        self.n = 0

    def before_run(self, local_flags) -> None:
        """In front of the kernel that WRITES `local_flags`: the current stream waits for any packing still reading that tensor."""
        import torch
        cur = torch.cuda.current_stream(self.device)
        for p in self.packed:
            if p is not None and p[1] == local_flags.data_ptr() and not p[0].query():    # (a wait costs the stream a packet: only if it is needed)
                cur.wait_event(p[0])

    def submit(self, local_flags) -> int:
        """Queue the exchange of this batch's flags (written by work already queued on the current stream)."""
        import torch
        import torch.distributed as dist
        k = self.n & 1
        self.n += 1
        cur = torch.cuda.current_stream(self.device)
        computed = torch.cuda.Event()
        computed.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(computed)           # (set k's previous exchange is earlier work of this same stream)
            pack_flags(local_flags, out=self.bits[k])
            packed = torch.cuda.Event()
            packed.record(self.side)
            dist.all_gather_into_tensor(self.gbits[k], self.bits[k], group=self.group)
            unpack_flags(self.gbits[k], self.E, out=self.out[k])
            done = torch.cuda.Event()
            done.record(self.side)
        self.packed[k] = (packed, local_flags.data_ptr())
        self.done[k] = done
        return k

    def result(self, k: int):
        """[total_channels, E] uint8 flags of exchange k, valid for work queued on the current stream after this call
        (and until the exchange after next reuses the set)."""
        import torch
        torch.cuda.current_stream(self.device).wait_event(self.done[k])
        return self.out[k]

    def synchronize(self):
        self.side.synchronize()


class HostFlagGather:
    """PipelinedFlagGather's interface over a CPU group (`gloo`): the bit-packed rows go to the host, through the group's
    all-gather, and back -- what a job falls back to when RCCL does not come up between its ranks (bench.py agrees on that over
    the gloo group before anything is timed).  Synchronous: the exchange of a batch is complete when submit() returns."""

    def __init__(self, local_rows: int, E: int, total_channels: int, device, group=None):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(group)
        if local_rows * world != total_channels:
            raise ValueError("HostFlagGather needs equal shards (use gather_flags on CPU tensors for ragged ones)")
        self.E, self.group, self.device = int(E), group, torch.device(device)
        nb = (self.E + 7) // 8
        self.bits = torch.empty((local_rows, nb), dtype=torch.uint8, device=self.device)
        self.hbits = torch.empty((local_rows, nb), dtype=torch.uint8).pin_memory()
        self.hall = torch.empty((total_channels, nb), dtype=torch.uint8).pin_memory()
        self.gbits = torch.empty((total_channels, nb), dtype=torch.uint8, device=self.device)
        self.out = [torch.empty((total_channels, self.E), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.n = 0

    def before_run(self, local_flags) -> None:
        pass                                             # (submit() has waited for the packing before it returned)

    def submit(self, local_flags) -> int:
        import torch
        import torch.distributed as dist
        k = self.n & 1
        self.n += 1
        pack_flags(local_flags, out=self.bits)
        self.hbits.copy_(self.bits)                      # (synchronous for the host: the kernel and the packing have finished)
        torch.cuda.current_stream(self.device).synchronize()
        dist.all_gather_into_tensor(self.hall, self.hbits, group=self.group)
        self.gbits.copy_(self.hall, non_blocking=True)
        unpack_flags(self.gbits, self.E, out=self.out[k])
        return k

    def result(self, k: int):
        return self.out[k]

    def synchronize(self):
        import torch
        torch.cuda.current_stream(self.device).synchronize()


class ShardedSyllableDetector:
    """This rank's share of a `total_channels`-wide bank on its own GPU: a block of channels, or -- with fewer channels than
    ranks -- a stretch of one channel's time axis (`time_sharded`; see the module docstring)."""

    def __init__(self, config, total_channels: int, device: Optional[int] = None, group=None, engine: int = 0):
        import torch.distributed as dist
        from .detector import SyllableDetector
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_channels = int(total_channels)
        self.first, self.count, self.part, self.parts = shard_plane(self.total_channels, self.world, self.rank)
        self.time_sharded = self.parts > 1
        if device is None:
            # one process per GPU: under torch.distributed.run every rank sees every GPU and owns the one of its LOCAL_RANK
            import os
            import torch
            device = int(os.environ["LOCAL_RANK"]) if "LOCAL_RANK" in os.environ else torch.cuda.current_device()
        self.device = int(device)
        self.detector = SyllableDetector(config, channels=self.count, device=self.device, engine=engine)
        self._time_range = int(config.timeRange)
        self._window = int(config.windowLength)

    def sample_range(self, n_samples: int) -> Tuple[int, int]:
        """[s0, s1) of a recording of n_samples per channel that this rank reads (all of it unless time-sharded)."""
        if not self.time_sharded:
            return 0, int(n_samples)
        g = self.detector.geometry
        e0, n = shard_evaluations(self.detector.countEvaluations(n_samples), self.parts, self.part)
        return time_shard_samples(g.hop, g.gap, self._window, self._time_range, e0, n)

    def run(self, local_samples, gather: bool = True, n_samples: Optional[int] = None):
        """local_samples: this rank's [count, S] block -- when time-sharded, its [1, s1 - s0] slice of a recording of
        `n_samples` (sample_range()).  Returns (outputs_local, flags) where flags is the gathered [total_channels, E]
        tensor when `gather`, else the local one."""
        if self.time_sharded:
            if n_samples is None:
                raise ValueError("a time-sharded run needs the recording's length (n_samples)")
            # A rank whose slice is not its sample_range() would otherwise raise alone while its peers wait in the gather for
            # ever: the check comes BEFORE anything is launched, and its verdict is agreed on by every rank (one tiny
            # all-reduce), so that either all ranks raise or none does.
            s0, s1 = self.sample_range(n_samples)
            ok = int(local_samples.shape[1]) == s1 - s0
            if gather and self.world > 1:
                import torch
                import torch.distributed as dist
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=local_samples.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
                all_ok = bool(int(t.item()))
            else:
                all_ok = ok
            if not all_ok:
                raise ValueError("local samples are not this rank's sample_range() of the recording" if not ok else
                                 "another rank's samples are not its sample_range() of the recording")
            outputs, flags = self.detector.run(local_samples)
            E = self.detector.countEvaluations(n_samples)
            if gather and self.world > 1:
                flags = gather_time_shards(flags, self.total_channels, E, self.group)
            return outputs, flags
        outputs, flags = self.detector.run(local_samples)
        if gather and self.world > 1:
            flags = gather_flags(flags, self.total_channels, self.group)
        return outputs, flags

    def close(self):
        self.detector.close()
