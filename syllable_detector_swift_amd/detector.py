"""Host-side mirror of SyllableDetector over the C ABI.

Keeps the reference's surface (Common/SyllableDetector.swift): init(config:) :37,
appendAudioData :129, processNewValue :153, lastOutputs :26, lastDetected :27,
seenSyllable :220 -- per channel of a bank -- and adds the batch entry points the MI355X
engine is built around (whole recordings of many channels in one pass).  All arithmetic
happens in libsyldet's HIP kernels; torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _abi
from .config import SyllableDetectorConfig, SyllableDetectorError, check


def _torch():
    import torch
    return torch


class SyllableDetector:
    def __init__(self, config: SyllableDetectorConfig, channels: int = 1, device: int = 0,
                 engine: int = _abi.ENGINE_AUTO):
        self.config = config
        self.channels = int(channels)
        self.device = int(device)
        self._h = _abi.Handle()
        c, keep = config.to_abi()
        check(_abi.lib.syldet_create(C.byref(c), self.channels, self.device, int(engine), C.byref(self._h)))
        del keep                      # the library copied every array
        g = _abi.Geometry()
        check(_abi.lib.syldet_get_geometry(self._h, C.byref(g)))
        self.geometry = g

    @classmethod
    def borrowed(cls, bank, shard: int) -> "SyllableDetector":
        """Shard `shard`'s own bank of a ShardedSyllableDetectorBank as a SyllableDetector (timings, fix-up statistics, spot
        checks); it belongs to the sharded bank and is not destroyed with this object."""
        self = cls.__new__(cls)
        self.config = bank.config
        self.channels = int(bank.shards[shard].channels)
        self.device = int(bank.shards[shard].device)
        self._h = _abi.Handle(_abi.lib.syldet_sharded_bank(bank._h, int(shard)))
        self._borrowed = True
        g = _abi.Geometry()
        check(_abi.lib.syldet_get_geometry(self._h, C.byref(g)))
        self.geometry = g
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                _abi.lib.syldet_destroy(self._h)
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

    # ---- geometry -----------------------------------------------------------------
    def countFrames(self, n_samples: int) -> int:
        return int(_abi.lib.syldet_count_frames(self._h, int(n_samples)))

    def countEvaluations(self, n_samples: int) -> int:
        return int(_abi.lib.syldet_count_evals(self._h, int(n_samples)))

    # ---- the reference's streaming API, one detector per channel -------------------
    def appendAudioData(self, data, channel: int = 0) -> None:
        a = np.ascontiguousarray(data, dtype=np.float32)
        check(_abi.lib.syldet_append(self._h, channel, a.ctypes.data_as(_abi.c_float_p), a.size))

    def appendInterleavedData(self, data, fromChannels=None) -> None:
        """appendInterleavedData(_:withSamples:fromChannel:ofTotalChannels:) (CircularShortTimeFourierTransform.swift:203-217):
        `data` [frames, total channels]; fromChannels (one stream channel per bank channel) picks a subset of a wider stream."""
        if fromChannels is None:
            a = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, self.channels)
            check(_abi.lib.syldet_append_interleaved(self._h, a.ctypes.data_as(_abi.c_float_p), a.shape[0], self.channels))
            return
        a = np.ascontiguousarray(data, np.float32)
        src = np.ascontiguousarray(fromChannels, np.int32)
        if a.ndim != 2 or src.shape != (self.channels,):
            raise ValueError("data [frames, total channels] and one source channel per bank channel")
        check(_abi.lib.syldet_append_interleaved_channels(self._h, a.ctypes.data_as(_abi.c_float_p), a.shape[0], a.shape[1],
                                                          src.ctypes.data_as(_abi.C.POINTER(_abi.C.c_int32))))

    def processNewValue(self, channel: int = 0) -> bool:
        return check(_abi.lib.syldet_process_new_value(self._h, channel)) == 1

    def processAll(self) -> int:
        """Evaluates what every channel has pending in one device round trip (the consumer loop of
        Processor.swift:128-141 over all detectors); returns the evaluations queued.  The following
        processNewValue calls hand them out without touching the device."""
        n = C.c_int64(0)
        check(_abi.lib.syldet_process_all(self._h, C.byref(n)))
        return n.value

    def pendingEvaluations(self, channel: int = 0) -> int:
        return check(_abi.lib.syldet_pending_evaluations(self._h, channel))

    def lastOutputsFor(self, channel: int) -> List[float]:
        out = np.zeros(self.geometry.outputs, np.float32)
        check(_abi.lib.syldet_last_outputs(self._h, channel, out.ctypes.data_as(_abi.c_float_p)))
        return out.tolist()

    @property
    def lastOutputs(self) -> List[float]:
        return self.lastOutputsFor(0)

    def lastDetectedFor(self, channel: int) -> bool:
        return check(_abi.lib.syldet_last_detected(self._h, channel)) == 1

    @property
    def lastDetected(self) -> bool:
        return self.lastDetectedFor(0)

    def seenSyllable(self, channel: int = 0) -> bool:
        return check(_abi.lib.syldet_seen_syllable(self._h, channel)) == 1

    # ---- batch, device tensors ----------------------------------------------------
    def _stream_ptr(self, stream) -> int:
        torch = _torch()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return int(s.cuda_stream)

    def _check_samples(self, samples):
        torch = _torch()
        if not (samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 2):
            raise ValueError("samples must be a 2-D float32 CUDA tensor [channels, n_samples]")
        if samples.shape[0] != self.channels or samples.stride(1) != 1:
            raise ValueError("samples must have one contiguous row per channel")
        if samples.device.index != self.device:
            raise ValueError("samples live on a different device than the detector")

    def _check_results(self, outputs, flags, E, device):
        """Caller-supplied result tensors go to the kernels as raw pointers: shape, dtype, layout and device must be right."""
        torch = _torch()
        if outputs is not None:
            want = (self.channels, E, self.geometry.outputs)
            if not (outputs.is_cuda and outputs.dtype == torch.float32 and tuple(outputs.shape) == want and outputs.is_contiguous()
                    and outputs.device == device):
                raise ValueError("outputs must be a contiguous float32 CUDA tensor of shape %s on the samples' device" % (want,))
        if flags is not None:
            want = (self.channels, E)
            if not (flags.is_cuda and flags.dtype == torch.uint8 and tuple(flags.shape) == want and flags.is_contiguous()
                    and flags.device == device):
                raise ValueError("flags must be a contiguous uint8 CUDA tensor of shape %s on the samples' device" % (want,))

    def run(self, samples, outputs=None, flags=None, stream=None):
        """samples [C, S] -> (outputs [C, E, n_out] f32, flags [C, E] u8), asynchronous on `stream`."""
        torch = _torch()
        self._check_samples(samples)
        S = int(samples.shape[1])
        E = self.countEvaluations(S)
        self._check_results(outputs, flags, E, samples.device)
        if outputs is None:
            outputs = torch.empty((self.channels, E, self.geometry.outputs), dtype=torch.float32, device=samples.device)
        if flags is None:
            flags = torch.empty((self.channels, E), dtype=torch.uint8, device=samples.device)
        check(_abi.lib.syldet_run_device(self._h, samples.data_ptr(), S, int(samples.stride(0)),
                                         outputs.data_ptr(), flags.data_ptr(), self._stream_ptr(stream)))
        return outputs, flags

    def runInterleaved(self, frames, outputs=None, flags=None, stream=None):
        """frames [n, C] (frame-major, as a decoder delivers audio) -> (outputs, flags) like run()."""
        torch = _torch()
        if not (frames.is_cuda and frames.dtype == torch.float32 and frames.dim() == 2 and frames.is_contiguous()):
            raise ValueError("frames must be a contiguous 2-D float32 CUDA tensor [n_frames, channels]")
        if frames.shape[1] != self.channels or frames.device.index != self.device:
            raise ValueError("frames must have one column per channel and live on the detector's device")
        n = int(frames.shape[0])
        E = max(self.countEvaluations(n), 0)
        self._check_results(outputs, flags, E, frames.device)
        if outputs is None:
            outputs = torch.empty((self.channels, E, self.geometry.outputs), dtype=torch.float32, device=frames.device)
        if flags is None:
            flags = torch.empty((self.channels, E), dtype=torch.uint8, device=frames.device)
        check(_abi.lib.syldet_run_interleaved_device(self._h, frames.data_ptr(), n, self.channels, outputs.data_ptr(),
                                                     flags.data_ptr(), self._stream_ptr(stream)))
        return outputs, flags

    def runInterleavedHost(self, frames: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        a = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, self.channels)
        n = a.shape[0]
        E = max(self.countEvaluations(n), 0)
        out = np.zeros((self.channels, E, self.geometry.outputs), np.float32)
        fl = np.zeros((self.channels, E), np.uint8)
        check(_abi.lib.syldet_run_interleaved(self._h, a.ctypes.data_as(_abi.c_float_p), n, self.channels,
                                              out.ctypes.data_as(_abi.c_float_p), fl.ctypes.data_as(_abi.c_uint8_p)))
        return out, fl

    def spectrogram(self, samples, stream=None):
        """samples [C, S] -> columns [C, J, bins] f32 (what processFourierData appends)."""
        torch = _torch()
        self._check_samples(samples)
        S = int(samples.shape[1])
        J = self.countFrames(S)
        cols = torch.empty((self.channels, J, self.geometry.bins), dtype=torch.float32, device=samples.device)
        check(_abi.lib.syldet_spectrogram_device(self._h, samples.data_ptr(), S, int(samples.stride(0)),
                                                 cols.data_ptr(), self._stream_ptr(stream)))
        return cols

    def detections(self, flags, debounce: float = 0.0, capacity: Optional[int] = None, stream=None):
        """flags [C, E] u8 -> (indices [C, capacity] i64, counts [C] i64); TrackDetector.swift:65-100."""
        torch = _torch()
        if not (flags.is_cuda and flags.dtype == torch.uint8 and flags.dim() == 2 and flags.shape[0] == self.channels
                and flags.is_contiguous() and flags.device.index == self.device):
            raise ValueError("flags must be a contiguous uint8 CUDA tensor [channels, n_evals] on the detector's device")
        E = int(flags.shape[1])
        cap = E if capacity is None else int(capacity)
        idx = torch.empty((self.channels, max(cap, 1)), dtype=torch.int64, device=flags.device)
        cnt = torch.empty((self.channels,), dtype=torch.int64, device=flags.device)
        check(_abi.lib.syldet_detections_device(self._h, flags.data_ptr(), E, float(debounce), idx.data_ptr(), cap,
                                                cnt.data_ptr(), self._stream_ptr(stream)))
        return idx, cnt

    # ---- measurement ------------------------------------------------------------------
    def profile(self, enable: bool = True, history: int = 1) -> None:
        """Bracket every kernel of a batch call with HIP events; `history`: how many calls' events to keep (a timing loop
        that reads them at its end need not wait for every call before making the next)."""
        check(_abi.lib.syldet_profile_history(self._h, int(history)))
        check(_abi.lib.syldet_profile(self._h, 1 if enable else 0))

    def timingsOf(self, calls_back: int = 0):
        """[(kernel name, milliseconds)] of the batch call made `calls_back` calls before the last one."""
        ms = (C.c_double * 8)()
        names = (C.c_char_p * 8)()
        n = C.c_int32()
        check(_abi.lib.syldet_timings(self._h, int(calls_back), ms, names, 8, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, 8))]

    def lastTimings(self):
        """[(kernel name, milliseconds)] of the last batch call (HIP events on its stream)."""
        ms = (C.c_double * 8)()
        names = (C.c_char_p * 8)()
        n = C.c_int32()
        check(_abi.lib.syldet_last_timings(self._h, ms, names, 8, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, 8))]

    def fixupStats(self):
        """(work items of 16 evaluations the last batch call recomputed exactly, overflow flag): the fused kernels' precision
        guard at work (0 for ordinary audio).  Synchronise the call's stream first."""
        items, over = C.c_int64(), C.c_int32()
        check(_abi.lib.syldet_fixup_stats(self._h, C.byref(items), C.byref(over)))
        return int(items.value), int(over.value)

    def segmentEvaluations(self, n_samples: int) -> int:
        """Evaluations per workgroup segment of the fused kernels for a batch of this length (0: no seams)."""
        return int(_abi.lib.syldet_segment_evals(self._h, int(n_samples)))

    # ---- batch, host arrays -------------------------------------------------------
    def runHost(self, samples: np.ndarray, outputs: Optional[np.ndarray] = None, flags: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """syldet_run: the batch call on host arrays, pipelined along time inside the library.  `outputs` / `flags`: arrays to
        write into (e.g. bank.PinnedArray views, which the DMA engines write in place)."""
        a = np.ascontiguousarray(samples, dtype=np.float32).reshape(self.channels, -1)
        S = a.shape[1]
        E = max(self.countEvaluations(S), 0)
        out = outputs if outputs is not None else np.zeros((self.channels, E, self.geometry.outputs), np.float32)
        fl = flags if flags is not None else np.zeros((self.channels, E), np.uint8)
        if out.shape != (self.channels, E, self.geometry.outputs) or out.dtype != np.float32 or not out.flags.c_contiguous:
            raise ValueError("outputs must be a C-contiguous float32 array [channels, E, outputs]")
        if fl.shape != (self.channels, E) or fl.dtype != np.uint8 or not fl.flags.c_contiguous:
            raise ValueError("flags must be a C-contiguous uint8 array [channels, E]")
        check(_abi.lib.syldet_run(self._h, a.ctypes.data_as(_abi.c_float_p), S, S,
                                  out.ctypes.data_as(_abi.c_float_p), fl.ctypes.data_as(_abi.c_uint8_p)))
        return out, fl

    def spectrogramHost(self, samples: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(samples, dtype=np.float32).reshape(self.channels, -1)
        S = a.shape[1]
        J = self.countFrames(S)
        cols = np.zeros((self.channels, J, self.geometry.bins), np.float32)
        check(_abi.lib.syldet_spectrogram(self._h, a.ctypes.data_as(_abi.c_float_p), S, S,
                                          cols.ctypes.data_as(_abi.c_float_p)))
        return cols

    def detectionsHost(self, flags: np.ndarray, debounce: float = 0.0, capacity: Optional[int] = None):
        f = np.ascontiguousarray(flags, dtype=np.uint8).reshape(self.channels, -1)
        E = f.shape[1]
        cap = E if capacity is None else int(capacity)
        idx = np.zeros((self.channels, max(cap, 1)), np.int64)
        cnt = np.zeros((self.channels,), np.int64)
        check(_abi.lib.syldet_detections(self._h, f.ctypes.data_as(_abi.c_uint8_p), E, float(debounce),
                                         idx.ctypes.data_as(_abi.c_int64_p), cap, cnt.ctypes.data_as(_abi.c_int64_p)))
        return idx, cnt
