"""Host-side mirror of a detector bank that spans several GPUs of ONE process (syldet_create_sharded).

The reference is one process that owns every channel -- Processor.swift:57-59 builds one SyllableDetector per channel and
one serial queue drains them (:82, :128-141); main.swift:86-89, :126-130 does the same per track -- so a host with several
MI355X keeps one handle and one call per batch: the library places a sub-bank and a stream on every listed device, splits
the channels into contiguous blocks (time-axis ranges when there are fewer channels than devices) and exchanges the detection
flags with ONE all-gather per batch (RCCL communicators made inside the library, ncclCommInitAll).  torch is used for
device memory only; `dist.py` remains the process-per-GPU form of the same sharding.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from .config import SyllableDetectorConfig, check


def shard_table(total_channels: int, n_shards: int) -> List[Tuple[int, int, int, int]]:
    """[(first channel, channels, part, parts)] per shard, from the library's own host arithmetic (no device touched)."""
    out = (_abi.Shard * int(n_shards))()
    check(_abi.lib.syldet_shard_table(int(total_channels), int(n_shards), out))
    return [(s.first_channel, s.channels, s.part, s.parts) for s in out]


class ShardedSyllableDetectorBank:
    def __init__(self, config: SyllableDetectorConfig, channels: int, devices: Sequence[int], engine: int = _abi.ENGINE_AUTO,
                 exchange: int = _abi.EXCHANGE_RCCL):
        self.config = config
        self.channels = int(channels)
        self.devices = [int(d) for d in devices]
        self._h = _abi.Handle()
        c, keep = config.to_abi()
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        check(_abi.lib.syldet_create_sharded(C.byref(c), self.channels, devs, len(self.devices), int(engine), int(exchange), C.byref(self._h)))
        del keep
        self.shards = []
        for i in range(len(self.devices)):
            s = _abi.Shard()
            check(_abi.lib.syldet_sharded_shard(self._h, i, C.byref(s)))
            self.shards.append(s)
        g = _abi.Geometry()
        check(_abi.lib.syldet_get_geometry(_abi.lib.syldet_sharded_bank(self._h, 0), C.byref(g)))
        self.geometry = g

    def close(self):
        if getattr(self, "_h", None):
            _abi.lib.syldet_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def rcclRanks(self) -> int:
        return int(_abi.lib.syldet_sharded_rccl_ranks(self._h))

    @property
    def launcherThreads(self) -> int:
        """Persistent per-shard launcher threads of the bank (0: one shard, or SYLDET_SHARDED_INLINE=1)."""
        return int(_abi.lib.syldet_sharded_launcher_threads(self._h))

    def connect(self) -> None:
        """Bring the exchange up now (RCCL: librccl + ncclCommInitAll; copy exchange: peer access) instead of inside the first
        gathering batch; raises on failure, so a caller can make the bank again with EXCHANGE_PEER_COPY in the same process."""
        check(_abi.lib.syldet_sharded_connect(self._h))

    def countEvaluations(self, n_samples: int) -> int:
        return int(_abi.lib.syldet_count_evals(_abi.lib.syldet_sharded_bank(self._h, 0), int(n_samples)))

    def countFrames(self, n_samples: int) -> int:
        return int(_abi.lib.syldet_count_frames(_abi.lib.syldet_sharded_bank(self._h, 0), int(n_samples)))

    def ranges(self, shard: int, n_samples: int) -> Tuple[int, int, int, int]:
        """(s0, s1, e0, count): the samples shard `shard` reads of each of its channels and the evaluations it computes."""
        v = [C.c_int64() for _ in range(4)]
        check(_abi.lib.syldet_sharded_ranges(self._h, int(shard), int(n_samples), *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    # ---- host arrays: the whole bank in one call ----------------------------------------------------------------
    def runHost(self, samples: np.ndarray, outputs: Optional[np.ndarray] = None, flags: Optional[np.ndarray] = None):
        a = np.ascontiguousarray(samples, dtype=np.float32).reshape(self.channels, -1) if not _is_rows(samples, self.channels) else samples
        S = a.shape[1]
        E = max(self.countEvaluations(S), 0)
        out = outputs if outputs is not None else np.zeros((self.channels, E, self.geometry.outputs), np.float32)
        fl = flags if flags is not None else np.zeros((self.channels, E), np.uint8)
        # (the library writes these rows from one thread per shard: a wrong array would be written out of bounds)
        if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.shape == (self.channels, E, self.geometry.outputs)):
            raise ValueError("outputs must be a C-contiguous float32 array [%d, %d, %d]" % (self.channels, E, self.geometry.outputs))
        if not (isinstance(fl, np.ndarray) and fl.dtype == np.uint8 and fl.flags["C_CONTIGUOUS"] and fl.shape == (self.channels, E)):
            raise ValueError("flags must be a C-contiguous uint8 array [%d, %d]" % (self.channels, E))
        check(_abi.lib.syldet_sharded_run(self._h, a.ctypes.data_as(_abi.c_float_p), S, a.strides[0] // 4,
                                          out.ctypes.data_as(_abi.c_float_p), fl.ctypes.data_as(_abi.c_uint8_p)))
        return out, fl

    # ---- device tensors: one block per shard ----------------------------------------------------------------------
    def scatter(self, samples_host: np.ndarray):
        """A whole recording [C, S] on the host -> the per-shard device blocks run() takes (test / bench convenience)."""
        import torch
        S = samples_host.shape[1]
        blocks = []
        for i, s in enumerate(self.shards):
            s0, s1, _, _ = self.ranges(i, S)
            rows = samples_host[s.first_channel: s.first_channel + s.channels, s0:s1]
            blocks.append(torch.from_numpy(np.ascontiguousarray(rows)).to(torch.device("cuda", s.device)))
        return blocks

    def run(self, blocks, n_samples: int, gather: bool = True, outputs=None, flags=None, flags_all=None):
        """blocks[i]: shard i's [channels_i, s1 - s0] float32 tensor on its device (ranges()).  Returns (outputs per shard,
        flags per shard, gathered [C, E] flags per device or None).  Asynchronous: synchronize() before reading."""
        import torch
        n = len(self.shards)
        E = max(self.countEvaluations(n_samples), 0)
        n_out = self.geometry.outputs
        outs, fls, alls = [], [], []
        for i, s in enumerate(self.shards):
            s0, s1, _, cnt = self.ranges(i, n_samples)
            x = blocks[i]
            dev = torch.device("cuda", s.device)
            if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.device == dev
                    and x.shape[0] == s.channels and x.shape[1] == s1 - s0):
                raise ValueError("block %d must be a float32 CUDA tensor [%d, %d] on device %d" % (i, s.channels, s1 - s0, s.device))
            outs.append(outputs[i] if outputs is not None else torch.empty((s.channels, cnt, n_out), dtype=torch.float32, device=dev))
            fls.append(flags[i] if flags is not None else torch.empty((s.channels, cnt), dtype=torch.uint8, device=dev))
            if gather:
                alls.append(flags_all[i] if flags_all is not None else torch.empty((self.channels, E), dtype=torch.uint8, device=dev))
            if (tuple(outs[-1].shape) != (s.channels, cnt, n_out) or tuple(fls[-1].shape) != (s.channels, cnt) or not outs[-1].is_contiguous()
                    or not fls[-1].is_contiguous() or outs[-1].dtype != torch.float32 or fls[-1].dtype != torch.uint8
                    or outs[-1].device != dev or fls[-1].device != dev):
                raise ValueError("result tensors of shard %d: float32 [%d, %d, %d] and uint8 [%d, %d], contiguous, on device %d" % (
                    i, s.channels, cnt, n_out, s.channels, cnt, s.device))
            if gather and (tuple(alls[-1].shape) != (self.channels, E) or not alls[-1].is_contiguous() or alls[-1].device != dev
                           or alls[-1].dtype != torch.uint8):
                raise ValueError("gathered flags of shard %d: uint8 [%d, %d], contiguous, on device %d" % (i, self.channels, E, s.device))
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        strides = (C.c_int64 * n)(*[int(b.stride(0)) for b in blocks])
        check(_abi.lib.syldet_sharded_run_device(self._h, arr(blocks), int(n_samples), strides, arr(outs), arr(fls),
                                                 arr(alls) if gather else None))
        return outs, fls, (alls if gather else None)

    def prepare(self, blocks, n_samples: int, outputs, flags, flags_all=None):
        """run()'s argument checks once, then a callable that queues one batch on exactly these tensors with nothing but the
        ABI call (a stream of batches over the same buffers: eight shards' worth of Python checks cost more than the library's
        queueing).  The callable keeps the tensors alive."""
        gather = flags_all is not None
        n = len(self.shards)
        import torch
        E = max(self.countEvaluations(n_samples), 0)
        n_out = self.geometry.outputs
        for i, s in enumerate(self.shards):
            s0, s1, _, cnt = self.ranges(i, n_samples)
            dev = torch.device("cuda", s.device)
            want = [(blocks[i], torch.float32, (s.channels, s1 - s0)), (outputs[i], torch.float32, (s.channels, cnt, n_out)),
                    (flags[i], torch.uint8, (s.channels, cnt))] + ([(flags_all[i], torch.uint8, (self.channels, E))] if gather else [])
            for t, dt, shape in want:
                if not (t.is_cuda and t.device == dev and t.dtype == dt and tuple(t.shape) == shape and (t.is_contiguous() or (t is blocks[i] and t.stride(1) == 1))):
                    raise ValueError("shard %d: a %s tensor %s on device %d is expected" % (i, dt, list(shape), s.device))
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        a_blocks, a_outs, a_fls, a_alls = arr(blocks), arr(outputs), arr(flags), (arr(flags_all) if gather else None)
        strides = (C.c_int64 * n)(*[int(b.stride(0)) for b in blocks])
        keep = (list(blocks), list(outputs), list(flags), list(flags_all) if gather else None)
        h, f, S = self._h, _abi.lib.syldet_sharded_run_device, int(n_samples)

        def call(_keep=keep):
            st = f(h, a_blocks, S, strides, a_outs, a_fls, a_alls)
            if st:
                check(st)
        return call

    def synchronize(self):
        check(_abi.lib.syldet_sharded_synchronize(self._h))

    def streams(self, shard: int) -> Tuple[int, int]:
        """(compute stream, exchange stream) of a shard as hipStream_t values: the shard's own results are ordered on the
        first, its gathered flags on the second."""
        return int(_abi.lib.syldet_sharded_stream(self._h, int(shard)) or 0), int(_abi.lib.syldet_sharded_exchange_stream(self._h, int(shard)) or 0)


def _is_rows(a, channels) -> bool:
    return isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[0] == channels and a.strides[1] == 4


class PinnedArray:
    """A numpy view of page-locked host memory from syldet_host_alloc: audio and result buffers the DMA engines read and
    write in place (syldet_run then skips its staging copies)."""

    def __init__(self, shape, dtype):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        n = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(_abi.lib.syldet_host_alloc(max(n, 1), C.byref(self._p)))
        buf = (C.c_char * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if getattr(self, "_p", None) and self._p.value:
            self.array = None
            _abi.lib.syldet_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
