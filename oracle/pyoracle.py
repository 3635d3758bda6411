"""ctypes front-end of the CPU oracle (oracle/syldet_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see syldet_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.

Also holds an independent, pure-Python reader of the reference's `key = value` network
format (SyllableDetectorConfig.swift:170-277) so the product's C++ parser can be checked
against a second implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libsyldet_oracle.so")

F32, F64 = 32, 64
RULE_FIRST, RULE_ANY = 0, 1
WIN = {"none": 0, "hamming": 1, "hanning": 2, "blackman": 3}
SCALE = {"linear": 0, "log": 1, "db": 2}
FN = {"l2normalize": 0, "normalize": 1, "normalizestd": 2, "mapminmax": 3, "mapstd": 4}
TF = {"TanSig": 0, "LogSig": 1, "PureLin": 2, "SatLin": 3}
MAX_FNS, MAX_LAYERS = 8, 8

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)


class _Fn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("count", C.c_int32), ("xoff", _f32p), ("gain", _f32p), ("y", C.c_float)]


class _Layer(C.Structure):
    _fields_ = [("inputs", C.c_int32), ("outputs", C.c_int32), ("transfer", C.c_int32),
                ("weights", _f32p), ("biases", _f32p)]


class _Config(C.Structure):
    _fields_ = [("sampling_rate", C.c_double), ("fourier_length", C.c_int32), ("window_length", C.c_int32),
                ("window_overlap", C.c_int32), ("freq_lo", C.c_double), ("freq_hi", C.c_double),
                ("time_range", C.c_int32), ("scaling", C.c_int32), ("window", C.c_int32), ("power_mode", C.c_int32),
                ("n_in_fns", C.c_int32), ("in_fns", _Fn * MAX_FNS),
                ("n_layers", C.c_int32), ("layers", _Layer * MAX_LAYERS),
                ("n_out_fns", C.c_int32), ("out_fns", _Fn * MAX_FNS),
                ("n_thresholds", C.c_int32), ("thresholds", _f64p)]


class _Geom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("gap", "overlap", "hop", "f0", "f1", "F", "I", "n_out")]


class _Resampler(C.Structure):
    _fields_ = [("step", C.c_float), ("last", C.c_float), ("offset", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                             for f in ("syldet_oracle.c", "syldet_oracle.h", "Makefile")):
        subprocess.run(["make", "-C", _HERE, "CC=gcc"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def build_native() -> str:
    """The same source at -O3 -march=native, built on the machine that runs it (the timed CPU baseline of bench.py)."""
    path = os.path.join(os.path.dirname(LIB_PATH), "libsyldet_oracle_native.so")
    subprocess.run(["make", "-B", "-C", _HERE, "CC=gcc", "native"], check=True, stdout=subprocess.DEVNULL)
    return path


_lib = None
_lib_native = None


def lib(native: bool = False):
    global _lib, _lib_native
    if native:
        if _lib_native is None:
            _lib_native = _load(build_native())
        return _lib_native
    if _lib is None:
        # SYLDET_ORACLE_LIB: another build of the SAME source (the AddressSanitizer build of tests/sanitize, run under
        # LD_PRELOAD=libasan by tests/test_sanitizers.py)
        alt = os.environ.get("SYLDET_ORACLE_LIB")
        if alt:
            _lib = _load(alt)
            return _lib
        if not os.path.exists(LIB_PATH):
            build()
        _lib = _load(LIB_PATH)
    return _lib


def _load(path: str):
    L = C.CDLL(path)
    cp = C.POINTER(_Config)
    L.orc_geometry.argtypes = [cp, C.POINTER(_Geom)]
    L.orc_window.argtypes = [C.c_int, C.c_int, _f32p]
    L.orc_window.restype = None
    L.orc_frequency_index_range.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_count_frames.argtypes = [cp, C.c_int64]
    L.orc_count_frames.restype = C.c_int64
    L.orc_count_evals.argtypes = [cp, C.c_int64]
    L.orc_count_evals.restype = C.c_int64
    L.orc_stft_frame.argtypes = [cp, _f32p, C.c_int, _f64p]
    L.orc_spectrogram.argtypes = [cp, _f32p, C.c_int64, C.c_int, _f64p]
    L.orc_spectrogram.restype = C.c_int64
    L.orc_net_apply.argtypes = [cp, _f32p, C.c_int, _f64p]
    L.orc_run.argtypes = [cp, _f32p, C.c_int64, C.c_int, C.c_int, _f32p, _u8p, _f64p]
    L.orc_run.restype = C.c_int64
    L.orc_detections.argtypes = [cp, _u8p, C.c_int64, C.c_double, _i64p, C.c_int64]
    L.orc_detections.restype = C.c_int64
    L.orc_stream_create.argtypes = [cp, C.c_int]
    L.orc_stream_create.restype = C.c_void_p
    L.orc_stream_destroy.argtypes = [C.c_void_p]
    L.orc_stream_destroy.restype = None
    L.orc_stream_append.argtypes = [C.c_void_p, _f32p, C.c_int64]
    L.orc_stream_process_new_value.argtypes = [C.c_void_p]
    L.orc_stream_last_outputs.argtypes = [C.c_void_p, _f32p]
    L.orc_stream_last_outputs.restype = None
    L.orc_stream_last_detected.argtypes = [C.c_void_p]
    L.orc_stream_seen_syllable.argtypes = [C.c_void_p]
    L.orc_resampler_init.argtypes = [C.POINTER(_Resampler), C.c_double, C.c_double]
    L.orc_resampler_init.restype = None
    L.orc_resampler_count.argtypes = [C.POINTER(_Resampler), C.c_int64]
    L.orc_resampler_count.restype = C.c_int64
    L.orc_resampler_run.argtypes = [C.POINTER(_Resampler), _f32p, C.c_int64, _f32p]
    L.orc_resampler_run.restype = C.c_int64
    L.orc_stream_run.argtypes = [cp, C.c_int, _f32p, C.c_int64, C.c_int64, _f32p, _u8p]
    L.orc_stream_run.restype = C.c_int64
    return L


# ------------------------------------------------------------------ network description

def parse_text(text: str) -> Dict:
    """Independent reader of the text format (SyllableDetectorConfig.swift:170-277)."""
    data = {}
    for line in text.split("\n"):
        parts = [p for p in line.split("=") if p != ""]          # Swift split drops empty pieces
        if len(parts) == 2:
            data[parts[0].strip()] = parts[1].strip()

    def floats(key, dtype=np.float32):
        return np.array([dtype(p.strip()) for p in data[key].split(",") if p != ""], dtype=dtype)

    def fns(prefix):
        out = []
        for i in range(int(data[prefix + "Count"])):
            nm = "%s%d" % (prefix, i)
            f = {"function": data[nm + ".function"]}
            if f["function"] in ("mapminmax", "mapstd"):
                f["xOffsets"] = floats(nm + ".xOffsets")
                f["gains"] = floats(nm + ".gains")
                f["y"] = np.float32(data[nm + (".yMin" if f["function"] == "mapminmax" else ".yMean")])
            out.append(f)
        return out

    layers = []
    for i in range(int(data["layers"])):
        nm = "layer%d" % i
        layers.append({"inputs": int(data[nm + ".inputs"]), "outputs": int(data[nm + ".outputs"]),
                       "weights": floats(nm + ".weights"), "biases": floats(nm + ".biases"),
                       "transferFunction": data[nm + ".transferFunction"]})
    fr = floats("freqRange", np.float64)
    thr = floats("thresholds", np.float64) if "thresholds" in data else floats("threshold", np.float64)
    flen = int(data["fourierLength"])
    return {"samplingRate": float(data["samplingRate"]), "fourierLength": flen,
            "windowLength": int(data.get("windowLength", flen)), "windowOverlap": int(data["windowOverlap"]),
            "freqRange": (float(fr[0]), float(fr[1])), "timeRange": int(data["timeRange"]),
            "scaling": data["scaling"], "thresholds": thr, "window": "hamming", "power_mode": 0,
            "inputs": fns("processInputs"), "layers": layers, "outputs": fns("processOutputs")}


def from_config(cfg) -> Dict:
    """Plain-data copy of a product SyllableDetectorConfig (attribute access only)."""
    def fn(f):
        d = {"function": f.function}
        if f.function in ("mapminmax", "mapstd"):
            d.update(xOffsets=np.asarray(f.xOffsets, np.float32), gains=np.asarray(f.gains, np.float32), y=np.float32(f.y))
        return d
    win = {0: "none", 1: "hamming", 2: "hanning", 3: "blackman"}[int(cfg.window)]
    return {"samplingRate": float(cfg.samplingRate), "fourierLength": int(cfg.fourierLength),
            "windowLength": int(cfg.windowLength), "windowOverlap": int(cfg.windowOverlap),
            "freqRange": (float(cfg.freqRange[0]), float(cfg.freqRange[1])), "timeRange": int(cfg.timeRange),
            "scaling": cfg.spectrogramScaling, "thresholds": np.asarray(cfg.thresholds, np.float64),
            "window": win, "power_mode": int(cfg.spectrum),
            "inputs": [fn(f) for f in cfg.net.inputProcessing],
            "layers": [{"inputs": int(L.inputs), "outputs": int(L.outputs),
                        "weights": np.asarray(L.weights, np.float32).reshape(-1),
                        "biases": np.asarray(L.biases, np.float32), "transferFunction": L.transferFunction}
                       for L in cfg.net.layers],
            "outputs": [fn(f) for f in cfg.net.outputProcessing]}


class Oracle:
    """One configured reference detector on the CPU."""

    def __init__(self, net: Dict):
        self.net = net
        self._keep = []
        c = _Config()
        c.sampling_rate = net["samplingRate"]
        c.fourier_length, c.window_length, c.window_overlap = net["fourierLength"], net["windowLength"], net["windowOverlap"]
        c.freq_lo, c.freq_hi = net["freqRange"]
        c.time_range = net["timeRange"]
        c.scaling = SCALE[net["scaling"]]
        c.window = WIN[net.get("window", "hamming")]
        c.power_mode = int(net.get("power_mode", 0))

        def ptr(a):
            a = np.ascontiguousarray(a, np.float32)
            self._keep.append(a)
            return a.ctypes.data_as(_f32p)

        def fill(dst, src):
            for i, f in enumerate(src):
                dst[i].kind = FN[f["function"]]
                if f["function"] in ("mapminmax", "mapstd"):
                    dst[i].count = len(f["xOffsets"])
                    dst[i].xoff, dst[i].gain, dst[i].y = ptr(f["xOffsets"]), ptr(f["gains"]), float(f["y"])
        c.n_in_fns = len(net["inputs"])
        fill(c.in_fns, net["inputs"])
        c.n_out_fns = len(net["outputs"])
        fill(c.out_fns, net["outputs"])
        c.n_layers = len(net["layers"])
        for i, L in enumerate(net["layers"]):
            c.layers[i].inputs, c.layers[i].outputs = L["inputs"], L["outputs"]
            c.layers[i].transfer = TF[L["transferFunction"]]
            c.layers[i].weights, c.layers[i].biases = ptr(L["weights"]), ptr(L["biases"])
        thr = np.ascontiguousarray(net["thresholds"], np.float64)
        self._keep.append(thr)
        c.n_thresholds = thr.size
        c.thresholds = thr.ctypes.data_as(_f64p)
        self.c = c
        g = _Geom()
        st = lib().orc_geometry(C.byref(c), C.byref(g))
        if st != 0:
            raise ValueError("oracle: invalid configuration (%d)" % st)
        self.g = g
        self.n_out = g.n_out

    def count_frames(self, S: int) -> int:
        return int(lib().orc_count_frames(C.byref(self.c), S))

    def count_evals(self, S: int) -> int:
        return int(lib().orc_count_evals(C.byref(self.c), S))

    def window(self) -> np.ndarray:
        w = np.zeros(self.c.window_length, np.float32)
        lib().orc_window(self.c.window, self.c.window_length, w.ctypes.data_as(_f32p))
        return w

    def stft_frame(self, x: np.ndarray, precision: int = F64) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        assert x.size >= self.c.window_length
        out = np.zeros(self.c.fourier_length // 2, np.float64)
        lib().orc_stft_frame(C.byref(self.c), x.ctypes.data_as(_f32p), precision, out.ctypes.data_as(_f64p))
        return out

    def spectrogram(self, samples: np.ndarray, precision: int = F64) -> np.ndarray:
        s = np.ascontiguousarray(samples, np.float32)
        J = self.count_frames(s.size)
        cols = np.zeros((max(J, 0), self.g.F), np.float64)
        if J > 0:
            lib().orc_spectrogram(C.byref(self.c), s.ctypes.data_as(_f32p), s.size, precision, cols.ctypes.data_as(_f64p))
        return cols

    def net_apply(self, v: np.ndarray, precision: int = F64) -> np.ndarray:
        v = np.ascontiguousarray(v, np.float32)
        assert v.size == self.g.I
        out = np.zeros(self.n_out, np.float64)
        lib().orc_net_apply(C.byref(self.c), v.ctypes.data_as(_f32p), precision, out.ctypes.data_as(_f64p))
        return out

    def run(self, samples: np.ndarray, precision: int = F64, rule: int = RULE_FIRST):
        """-> (outputs f32 [E][n_out], flags u8 [E], outputs64 f64 [E][n_out])"""
        s = np.ascontiguousarray(samples, np.float32)
        E = max(self.count_evals(s.size), 0)
        out = np.zeros((E, self.n_out), np.float32)
        fl = np.zeros(E, np.uint8)
        o64 = np.zeros((E, self.n_out), np.float64)
        if E > 0:
            lib().orc_run(C.byref(self.c), s.ctypes.data_as(_f32p), s.size, precision, rule,
                          out.ctypes.data_as(_f32p), fl.ctypes.data_as(_u8p), o64.ctypes.data_as(_f64p))
        return out, fl, o64

    def stream_run(self, samples: np.ndarray, precision: int = F32, chunk: int = 8192, native: bool = False, keep: bool = True):
        """The reference's consumer loop, frame at a time through its two rings (orc_stream_run): -> (outputs f32, flags u8),
        or the number of evaluations when keep is False (timing)."""
        s = np.ascontiguousarray(samples, np.float32)
        E = max(self.count_evals(s.size), 0)
        out = np.zeros((E, self.n_out), np.float32) if keep else None
        fl = np.zeros(E, np.uint8) if keep else None
        n = lib(native).orc_stream_run(C.byref(self.c), precision, s.ctypes.data_as(_f32p), s.size, chunk,
                                       out.ctypes.data_as(_f32p) if keep else None, fl.ctypes.data_as(_u8p) if keep else None)
        assert n == E, (n, E)
        return (out, fl) if keep else n

    def detections(self, flags: np.ndarray, debounce: float = 0.0) -> np.ndarray:
        f = np.ascontiguousarray(flags, np.uint8)
        idx = np.zeros(max(f.size, 1), np.int64)
        n = lib().orc_detections(C.byref(self.c), f.ctypes.data_as(_u8p), f.size, debounce,
                                 idx.ctypes.data_as(_i64p), idx.size)
        return idx[:n].copy()

    def stream(self, precision: int = F32) -> "OracleStream":
        return OracleStream(self, precision)


class OracleStream:
    """The reference's streaming object: appendAudioData / processNewValue / lastOutputs."""

    def __init__(self, o: Oracle, precision: int):
        self.o = o
        self.h = lib().orc_stream_create(C.byref(o.c), precision)
        if not self.h:
            raise ValueError("oracle stream: invalid configuration")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_stream_destroy(self.h)
            self.h = None

    def append(self, data: np.ndarray) -> int:
        d = np.ascontiguousarray(data, np.float32)
        return lib().orc_stream_append(self.h, d.ctypes.data_as(_f32p), d.size)

    def process_new_value(self) -> bool:
        return lib().orc_stream_process_new_value(self.h) == 1

    def last_outputs(self) -> np.ndarray:
        out = np.zeros(self.o.n_out, np.float32)
        lib().orc_stream_last_outputs(self.h, out.ctypes.data_as(_f32p))
        return out

    def last_detected(self) -> bool:
        return lib().orc_stream_last_detected(self.h) == 1

    def seen_syllable(self) -> bool:
        return lib().orc_stream_seen_syllable(self.h) == 1


class Resampler:
    """ResamplerLinear (Common/Resampler.swift:20-76)."""

    def __init__(self, rate_in: float, rate_out: float):
        self.r = _Resampler()
        lib().orc_resampler_init(C.byref(self.r), rate_in, rate_out)

    def resample(self, data: np.ndarray) -> np.ndarray:
        d = np.ascontiguousarray(data, np.float32)
        n = lib().orc_resampler_count(C.byref(self.r), d.size)
        out = np.zeros(max(n, 0), np.float32)
        m = lib().orc_resampler_run(C.byref(self.r), d.ctypes.data_as(_f32p), d.size, out.ctypes.data_as(_f32p))
        return out[:m]
