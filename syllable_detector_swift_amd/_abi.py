"""ctypes declaration of the C ABI in include/syldet.h.

The product path is libsyldet.so (hand-written HIP for gfx950).  There is no Python or
CPU fallback: if the library is missing this module raises at import time.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SYLDET_LIB: a diagnostic build of the same library (knock-out timing runs); never a different implementation
LIB_PATH = os.environ.get("SYLDET_LIB") or os.path.join(_HERE, "lib", "libsyldet.so")

ABI_VERSION = 1

# status codes (syldet_status_t)
OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_FFT_SIZE = -2
ERR_OVERLAP = -3
ERR_FREQ_RANGE = -4
ERR_INPUT_MISMATCH = -5
ERR_THRESHOLD_MISMATCH = -6
ERR_LAYER_SHAPE = -7
ERR_BUFFER_FULL = -8
ERR_NO_DEVICE = -9
ERR_DEVICE = -10
ERR_OUT_OF_MEMORY = -11
ERR_PARSE_OPEN = -20
ERR_PARSE_MISSING = -21
ERR_PARSE_INVALID = -22
ERR_PARSE_LENGTH = -23
ERR_UNSUPPORTED = -30

WINDOW_NONE, WINDOW_HAMMING, WINDOW_HANNING, WINDOW_BLACKMAN = 0, 1, 2, 3
SCALING_LINEAR, SCALING_LOG, SCALING_DB = 0, 1, 2
SPECTRUM_POWER, SPECTRUM_MAGNITUDE = 0, 1
FN_L2NORMALIZE, FN_NORMALIZE, FN_NORMALIZESTD, FN_MAPMINMAX, FN_MAPSTD = 0, 1, 2, 3, 4
TF_TANSIG, TF_LOGSIG, TF_PURELIN, TF_SATLIN = 0, 1, 2, 3
RULE_FIRST, RULE_ANY = 0, 1
ENGINE_AUTO, ENGINE_GENERIC, ENGINE_FUSED, ENGINE_WIDE_BF16 = 0, 1, 2, 3

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_uint8_p = C.POINTER(C.c_uint8)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)


class Fn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("count", C.c_int32), ("x_offsets", c_float_p),
                ("gains", c_float_p), ("y", C.c_float)]


class Layer(C.Structure):
    _fields_ = [("inputs", C.c_int32), ("outputs", C.c_int32), ("transfer", C.c_int32),
                ("weights", c_float_p), ("biases", c_float_p)]


class Config(C.Structure):
    _fields_ = [("sampling_rate", C.c_double),
                ("fourier_length", C.c_int32), ("window_length", C.c_int32), ("window_overlap", C.c_int32),
                ("freq_lo", C.c_double), ("freq_hi", C.c_double),
                ("time_range", C.c_int32), ("scaling", C.c_int32), ("window", C.c_int32),
                ("spectrum", C.c_int32), ("rule", C.c_int32),
                ("n_input_fns", C.c_int32), ("input_fns", C.POINTER(Fn)),
                ("n_layers", C.c_int32), ("layers", C.POINTER(Layer)),
                ("n_output_fns", C.c_int32), ("output_fns", C.POINTER(Fn)),
                ("n_thresholds", C.c_int32), ("thresholds", c_double_p)]


class Geometry(C.Structure):
    _fields_ = [("gap", C.c_int32), ("overlap", C.c_int32), ("hop", C.c_int32),
                ("f0", C.c_int32), ("f1", C.c_int32), ("bins", C.c_int32),
                ("inputs", C.c_int32), ("outputs", C.c_int32), ("first_index", C.c_int32),
                ("engine", C.c_int32)]


class Shard(C.Structure):
    _fields_ = [("device", C.c_int32), ("first_channel", C.c_int32), ("channels", C.c_int32),
                ("part", C.c_int32), ("parts", C.c_int32)]


EXCHANGE_RCCL, EXCHANGE_PEER_COPY = 0, 1

Handle = C.c_void_p
Config_p = C.POINTER(Config)
c_void_pp = C.POINTER(C.c_void_p)

# every symbol include/syldet.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "syldet_abi_version": (C.c_int, []),
    "syldet_strerror": (C.c_char_p, [C.c_int]),
    "syldet_last_error": (C.c_char_p, []),
    "syldet_config_load_text": (C.c_int, [C.c_char_p, C.POINTER(Config_p)]),
    "syldet_config_free": (None, [Config_p]),
    "syldet_config_geometry": (C.c_int, [Config_p, C.POINTER(Geometry)]),
    "syldet_frequency_index_range": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double, c_int32_p, c_int32_p]),
    "syldet_make_window": (C.c_int, [C.c_int32, C.c_int32, c_float_p]),
    "syldet_create": (C.c_int, [Config_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Handle)]),
    "syldet_destroy": (C.c_int, [Handle]),
    "syldet_get_geometry": (C.c_int, [Handle, C.POINTER(Geometry)]),
    "syldet_channels": (C.c_int32, [Handle]),
    "syldet_count_frames": (C.c_int64, [Handle, C.c_int64]),
    "syldet_count_evals": (C.c_int64, [Handle, C.c_int64]),
    "syldet_run_device": (C.c_int, [Handle, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "syldet_run": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int64, c_float_p, c_uint8_p]),
    "syldet_spectrogram_device": (C.c_int, [Handle, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "syldet_spectrogram": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int64, c_float_p]),
    "syldet_detections_device": (C.c_int, [Handle, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "syldet_detections": (C.c_int, [Handle, c_uint8_p, C.c_int64, C.c_double, c_int64_p, C.c_int64, c_int64_p]),
    "syldet_profile": (C.c_int, [Handle, C.c_int]),
    "syldet_last_timings": (C.c_int, [Handle, c_double_p, C.POINTER(C.c_char_p), C.c_int32, c_int32_p]),
    "syldet_profile_history": (C.c_int, [Handle, C.c_int32]),
    "syldet_timings": (C.c_int, [Handle, C.c_int32, c_double_p, C.POINTER(C.c_char_p), C.c_int32, c_int32_p]),
    "syldet_fixup_stats": (C.c_int, [Handle, c_int64_p, c_int32_p]),
    "syldet_segment_evals": (C.c_int64, [Handle, C.c_int64]),
    "syldet_append": (C.c_int, [Handle, C.c_int32, c_float_p, C.c_int64]),
    "syldet_append_interleaved": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int32]),
    "syldet_append_interleaved_channels": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]),
    "syldet_process_new_value": (C.c_int, [Handle, C.c_int32]),
    "syldet_process_all": (C.c_int, [Handle, c_int64_p]),
    "syldet_pending_evaluations": (C.c_int64, [Handle, C.c_int32]),
    "syldet_last_outputs": (C.c_int, [Handle, C.c_int32, c_float_p]),
    "syldet_last_detected": (C.c_int, [Handle, C.c_int32]),
    "syldet_seen_syllable": (C.c_int, [Handle, C.c_int32]),
    "syldet_deinterleave_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "syldet_run_interleaved_device": (C.c_int, [Handle, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "syldet_run_interleaved": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int32, c_float_p, c_uint8_p]),
    "syldet_pack_flags_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "syldet_unpack_flags_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "syldet_resampler_create": (C.c_int, [C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(Handle)]),
    "syldet_resampler_destroy": (C.c_int, [Handle]),
    "syldet_resampler_count": (C.c_int64, [Handle, C.c_int64]),
    "syldet_resample_device": (C.c_int, [Handle, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, c_int64_p, C.c_void_p]),
    "syldet_convert_rate_count": (C.c_int64, [C.c_int64, C.c_double, C.c_double]),
    "syldet_convert_rate_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int64, c_int64_p, C.c_void_p]),
    "syldet_resample": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int64, c_float_p, C.c_int64, c_int64_p]),
    "syldet_host_alloc": (C.c_int, [C.c_size_t, c_void_pp]),
    "syldet_host_free": (C.c_int, [C.c_void_p]),
    "syldet_shard_table": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(Shard)]),
    "syldet_shard_evaluations": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, c_int64_p, c_int64_p]),
    "syldet_shard_samples": (C.c_int, [Config_p, C.c_int64, C.c_int64, c_int64_p, c_int64_p]),
    "syldet_create_sharded": (C.c_int, [Config_p, C.c_int32, c_int32_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Handle)]),
    "syldet_sharded_destroy": (C.c_int, [Handle]),
    "syldet_sharded_channels": (C.c_int32, [Handle]),
    "syldet_sharded_shards": (C.c_int32, [Handle]),
    "syldet_sharded_shard": (C.c_int, [Handle, C.c_int32, C.POINTER(Shard)]),
    "syldet_sharded_bank": (Handle, [Handle, C.c_int32]),
    "syldet_sharded_stream": (C.c_void_p, [Handle, C.c_int32]),
    "syldet_sharded_exchange_stream": (C.c_void_p, [Handle, C.c_int32]),
    "syldet_sharded_ranges": (C.c_int, [Handle, C.c_int32, C.c_int64, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    "syldet_sharded_run": (C.c_int, [Handle, c_float_p, C.c_int64, C.c_int64, c_float_p, c_uint8_p]),
    "syldet_sharded_run_device": (C.c_int, [Handle, c_void_pp, C.c_int64, c_int64_p, c_void_pp, c_void_pp, c_void_pp]),
    "syldet_sharded_synchronize": (C.c_int, [Handle]),
    "syldet_sharded_rccl_ranks": (C.c_int32, [Handle]),
    "syldet_sharded_connect": (C.c_int, [Handle]),
    "syldet_sharded_launcher_threads": (C.c_int32, [Handle]),
}


def _share_hip_runtime_with_torch() -> None:
    """The torch wheel carries its own libamdhip64.so (SONAME libamdhip64.so.7) and asks for it by
    file name.  If libsyldet pulled /opt/rocm's copy in first, a later `import torch` would load a
    second HIP runtime into the process and one of the two would see no device.  Loading torch's
    copy by path first makes both sides resolve to the same runtime, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def load(path: str = LIB_PATH) -> C.CDLL:
    _share_hip_runtime_with_torch()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: libsyldet (the HIP/gfx950 engine) has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repository root. "
            "There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.syldet_abi_version() != ABI_VERSION:
        raise ImportError(f"libsyldet ABI {lib.syldet_abi_version()} != expected {ABI_VERSION}")
    return lib


lib = load()


def last_error() -> str:
    return (lib.syldet_last_error() or b"").decode("utf-8", "replace")


def strerror(status: int) -> str:
    return (lib.syldet_strerror(status) or b"").decode("utf-8", "replace")
