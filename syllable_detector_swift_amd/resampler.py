"""ResamplerLinear (Common/Resampler.swift:20-76) for a bank of channels on the GPU.

Same surface as the reference class -- `ResamplerLinear(fromRate:toRate:)`, `resampleVector`, `resampleArray`
-- over libsyldet's `syldet_resample*`; state (fractional offset, last sample of every channel) carries over
between calls exactly as in the reference.
"""
import ctypes as C

import numpy as np

from . import _abi
from .config import check


def deinterleave(frames, first_channel: int = 0, channels=None, stream=None):
    """frames [n, total] float32 CUDA tensor -> [channels, n] channel-major (appendInterleavedData's strided copy,
    CircularShortTimeFourierTransform.swift:203-217, for all requested channels at once)."""
    import torch
    if not (frames.is_cuda and frames.dtype == torch.float32 and frames.dim() == 2 and frames.is_contiguous()):
        raise ValueError("frames must be a contiguous 2-D float32 CUDA tensor [n_frames, total_channels]")
    n, total = int(frames.shape[0]), int(frames.shape[1])
    channels = total - first_channel if channels is None else int(channels)
    out = torch.empty((channels, n), dtype=torch.float32, device=frames.device)
    s = stream if stream is not None else torch.cuda.current_stream(frames.device)
    check(_abi.lib.syldet_deinterleave_device(frames.data_ptr(), n, total, int(first_channel), channels, out.data_ptr(), n,
                                              int(s.cuda_stream)))
    return out


class ResamplerLinear:
    def __init__(self, fromRate: float, toRate: float, channels: int = 1, device: int = 0):
        self.samplingRateIn, self.samplingRateOut = float(fromRate), float(toRate)
        self.channels, self.device = int(channels), int(device)
        h = _abi.Handle()
        check(_abi.lib.syldet_resampler_create(self.samplingRateIn, self.samplingRateOut, self.channels, self.device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _abi.lib.syldet_resampler_destroy(self._h)
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
        return False

    def countOutput(self, n_in: int) -> int:
        return int(_abi.lib.syldet_resampler_count(self._h, int(n_in)))

    def resampleVector(self, data, stream=None):
        """data [C, n] (or [n] for one channel) float32 CUDA tensor -> [C, n_out] resampled, asynchronous on `stream`."""
        import torch
        x = data if data.dim() == 2 else data.reshape(1, -1)
        if not (x.is_cuda and x.dtype == torch.float32 and x.shape[0] == self.channels and x.stride(1) == 1):
            raise ValueError("data must be a float32 CUDA tensor with one contiguous row per channel")
        n_in = int(x.shape[1])
        n_out = self.countOutput(n_in)
        out = torch.empty((self.channels, n_out), dtype=torch.float32, device=x.device)
        got = C.c_int64(0)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        check(_abi.lib.syldet_resample_device(self._h, x.data_ptr(), n_in, int(x.stride(0)), out.data_ptr(), max(n_out, 1),
                                              C.byref(got), int(s.cuda_stream)))
        assert got.value == n_out
        return out if data.dim() == 2 else out.reshape(-1)

    def resampleArray(self, arr) -> np.ndarray:
        """Host arrays ([n] or [C, n]); the reference's test helper (:71-75)."""
        a = np.ascontiguousarray(arr, dtype=np.float32)
        x = a.reshape(self.channels, -1)
        n_in = x.shape[1]
        n_out = self.countOutput(n_in)
        out = np.zeros((self.channels, n_out), np.float32)
        got = C.c_int64(0)
        check(_abi.lib.syldet_resample(self._h, x.ctypes.data_as(_abi.c_float_p), n_in, n_in,
                                       out.ctypes.data_as(_abi.c_float_p), max(n_out, 1), C.byref(got)))
        assert got.value == n_out
        return out if a.ndim == 2 else out.reshape(-1)
