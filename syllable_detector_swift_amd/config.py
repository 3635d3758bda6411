"""Host-side mirror of SyllableDetectorConfig / NeuralNet (configuration only).

Mirrors Common/SyllableDetectorConfig.swift:11-45 (stored fields, same names) and the
NeuralNet object graph of Common/NeuralNet.swift:231-378 as plain data.  Parsing of the
`key = value` text format is done by libsyldet (syldet_config_load_text, the C++
restatement of SyllableDetectorConfig.init(fromTextFile:), :170-277); this module only
moves the result in and out of the C structs.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi

_FN_NAMES = {_abi.FN_L2NORMALIZE: "l2normalize", _abi.FN_NORMALIZE: "normalize",
             _abi.FN_NORMALIZESTD: "normalizestd", _abi.FN_MAPMINMAX: "mapminmax", _abi.FN_MAPSTD: "mapstd"}
_FN_KINDS = {v: k for k, v in _FN_NAMES.items()}
_TF_NAMES = {_abi.TF_TANSIG: "TanSig", _abi.TF_LOGSIG: "LogSig", _abi.TF_PURELIN: "PureLin", _abi.TF_SATLIN: "SatLin"}
_TF_KINDS = {v: k for k, v in _TF_NAMES.items()}
_SCALING_NAMES = {_abi.SCALING_LINEAR: "linear", _abi.SCALING_LOG: "log", _abi.SCALING_DB: "db"}
_SCALING_KINDS = {v: k for k, v in _SCALING_NAMES.items()}


class SyllableDetectorError(RuntimeError):
    """A status the reference reports with fatalError (SyllableDetector.swift:47,54,59;
    CircularShortTimeFourierTransform.swift:77,83,87,199)."""

    def __init__(self, status: int, message: str = ""):
        self.status = status
        super().__init__(f"{_abi.strerror(status)} [{status}] {message}".strip())


class ParseError(SyllableDetectorError):
    """SyllableDetectorConfig.ParseError, SyllableDetectorConfig.swift:50-55."""
    kind = "parseError"


class UnableToOpenPath(ParseError):
    kind = "unableToOpenPath"


class MissingValue(ParseError):
    kind = "missingValue"


class InvalidValue(ParseError):
    kind = "invalidValue"


class MismatchedLength(ParseError):
    kind = "mismatchedLength"


_PARSE_ERRORS = {_abi.ERR_PARSE_OPEN: UnableToOpenPath, _abi.ERR_PARSE_MISSING: MissingValue,
                 _abi.ERR_PARSE_INVALID: InvalidValue, _abi.ERR_PARSE_LENGTH: MismatchedLength}


def check(status: int) -> int:
    """Raise for a negative status; pass through 0/1."""
    if status >= 0:
        return status
    msg = _abi.last_error()
    raise _PARSE_ERRORS.get(status, SyllableDetectorError)(status, msg)


@dataclass
class ProcessingFunction:
    """One input/output processing function (NeuralNet.swift:41-182)."""
    function: str                       # l2normalize | normalize | normalizestd | mapminmax | mapstd
    xOffsets: Optional[np.ndarray] = None
    gains: Optional[np.ndarray] = None
    y: float = 0.0                      # yMin (mapminmax) / yMean (mapstd)


@dataclass
class NeuralNetLayer:
    """NeuralNet.swift:329-378; weights row-major [outputs][inputs]."""
    inputs: int
    outputs: int
    weights: np.ndarray
    biases: np.ndarray
    transferFunction: str               # TanSig | LogSig | PureLin | SatLin


@dataclass
class NeuralNet:
    layers: List[NeuralNetLayer]
    inputProcessing: List[ProcessingFunction] = field(default_factory=list)
    outputProcessing: List[ProcessingFunction] = field(default_factory=list)

    @property
    def inputs(self) -> int:
        return self.layers[0].inputs

    @property
    def outputs(self) -> int:
        return self.layers[-1].outputs


@dataclass
class SyllableDetectorConfig:
    samplingRate: float
    fourierLength: int
    windowLength: int
    windowOverlap: int
    freqRange: Tuple[float, float]
    timeRange: int
    spectrogramScaling: str             # linear | log | db
    thresholds: List[float]
    net: NeuralNet
    # options the reference's detector fixes (hamming window SyllableDetector.swift:43,
    # extractPower :136, lastDetected on output 0 :27-31); exposed for the STFT class's
    # other modes and the CLI's any-output rule (TrackDetector.swift:72-77)
    window: int = _abi.WINDOW_HAMMING
    spectrum: int = _abi.SPECTRUM_POWER
    rule: int = _abi.RULE_FIRST

    # ---- text format -------------------------------------------------------------
    @classmethod
    def fromTextFile(cls, path: str) -> "SyllableDetectorConfig":
        """SyllableDetectorConfig.init(fromTextFile:) -- parsed by libsyldet."""
        p = _abi.Config_p()
        check(_abi.lib.syldet_config_load_text(str(path).encode(), C.byref(p)))
        try:
            return cls._from_abi(p.contents)
        finally:
            _abi.lib.syldet_config_free(p)

    @classmethod
    def _from_abi(cls, c: _abi.Config) -> "SyllableDetectorConfig":
        def arr(ptr, n, dtype):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True) if n > 0 else np.zeros(0, dtype)

        def fns(ptr, n):
            out = []
            for i in range(n):
                f = ptr[i]
                if f.count > 0:
                    out.append(ProcessingFunction(_FN_NAMES[f.kind], arr(f.x_offsets, f.count, np.float32),
                                                  arr(f.gains, f.count, np.float32), float(f.y)))
                else:
                    out.append(ProcessingFunction(_FN_NAMES[f.kind]))
            return out

        layers = []
        for i in range(c.n_layers):
            L = c.layers[i]
            layers.append(NeuralNetLayer(L.inputs, L.outputs,
                                         arr(L.weights, L.inputs * L.outputs, np.float32).reshape(L.outputs, L.inputs),
                                         arr(L.biases, L.outputs, np.float32), _TF_NAMES[L.transfer]))
        net = NeuralNet(layers, fns(c.input_fns, c.n_input_fns), fns(c.output_fns, c.n_output_fns))
        return cls(c.sampling_rate, c.fourier_length, c.window_length, c.window_overlap, (c.freq_lo, c.freq_hi),
                   c.time_range, _SCALING_NAMES[c.scaling], list(arr(c.thresholds, c.n_thresholds, np.float64)),
                   net, c.window, c.spectrum, c.rule)

    def toText(self) -> str:
        """Writes the format convert_to_text.m:61-212 emits (%.15g numbers)."""
        def num(v):
            return "%.15g" % float(v)

        def vec(a):
            return ", ".join(num(v) for v in np.asarray(a).reshape(-1))

        out = ["# AUTOMATICALLY GENERATED SYLLABLE DETECTOR CONFIGURATION",
               "samplingRate = %.1f" % self.samplingRate,
               "fourierLength = %d" % self.fourierLength,
               "windowLength = %d" % self.windowLength,
               "windowOverlap = %d" % self.windowOverlap,
               "freqRange = %.1f, %.1f" % tuple(self.freqRange),
               "timeRange = %d" % self.timeRange,
               "thresholds = " + vec(self.thresholds),
               "scaling = " + self.spectrogramScaling]
        for nm, fl in (("processInputs", self.net.inputProcessing), ("processOutputs", self.net.outputProcessing)):
            out.append("%sCount = %d" % (nm, len(fl)))
            for k, f in enumerate(fl):
                out.append("%s%d.function = %s" % (nm, k, f.function))
                if f.function in ("mapminmax", "mapstd"):
                    out.append("%s%d.xOffsets = %s" % (nm, k, vec(f.xOffsets)))
                    out.append("%s%d.gains = %s" % (nm, k, vec(f.gains)))
                    out.append("%s%d.%s = %s" % (nm, k, "yMin" if f.function == "mapminmax" else "yMean", num(f.y)))
        out.append("layers = %d" % len(self.net.layers))
        for i, L in enumerate(self.net.layers):
            out += ["layer%d.inputs = %d" % (i, L.inputs), "layer%d.outputs = %d" % (i, L.outputs),
                    "layer%d.weights = %s" % (i, vec(L.weights)), "layer%d.biases = %s" % (i, vec(L.biases)),
                    "layer%d.transferFunction = %s" % (i, L.transferFunction)]
        return "\n".join(out) + "\n"

    # ---- C struct -----------------------------------------------------------------
    def to_abi(self):
        """Returns (Config, keepalive): the struct points into `keepalive`'s arrays."""
        keep = []

        def fptr(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(_abi.c_float_p)

        def fn_array(fl: Sequence[ProcessingFunction]):
            arr = (_abi.Fn * max(len(fl), 1))()
            for i, f in enumerate(fl):
                arr[i].kind = _FN_KINDS[f.function]
                if f.function in ("mapminmax", "mapstd"):
                    if np.asarray(f.gains).size != np.asarray(f.xOffsets).size:     # the C side copies `count` of each
                        raise ValueError("%s: %d gains for %d xOffsets" % (f.function, np.asarray(f.gains).size, np.asarray(f.xOffsets).size))
                    arr[i].count = int(np.asarray(f.xOffsets).size)
                    arr[i].x_offsets = fptr(f.xOffsets)
                    arr[i].gains = fptr(f.gains)
                    arr[i].y = float(f.y)
            keep.append(arr)
            return arr

        c = _abi.Config()
        c.sampling_rate = float(self.samplingRate)
        c.fourier_length = int(self.fourierLength)
        c.window_length = int(self.windowLength)
        c.window_overlap = int(self.windowOverlap)
        c.freq_lo, c.freq_hi = float(self.freqRange[0]), float(self.freqRange[1])
        c.time_range = int(self.timeRange)
        c.scaling = _SCALING_KINDS[self.spectrogramScaling]
        c.window, c.spectrum, c.rule = int(self.window), int(self.spectrum), int(self.rule)
        c.n_input_fns = len(self.net.inputProcessing)
        c.input_fns = fn_array(self.net.inputProcessing)
        c.n_output_fns = len(self.net.outputProcessing)
        c.output_fns = fn_array(self.net.outputProcessing)
        layers = (_abi.Layer * max(len(self.net.layers), 1))()
        for i, L in enumerate(self.net.layers):
            layers[i].inputs, layers[i].outputs = int(L.inputs), int(L.outputs)
            layers[i].transfer = _TF_KINDS[L.transferFunction]
            layers[i].weights = fptr(np.asarray(L.weights).reshape(-1))
            layers[i].biases = fptr(L.biases)
        keep.append(layers)
        c.n_layers = len(self.net.layers)
        c.layers = layers
        thr = np.ascontiguousarray(self.thresholds, dtype=np.float64)
        keep.append(thr)
        c.n_thresholds = int(thr.size)
        c.thresholds = thr.ctypes.data_as(_abi.c_double_p)
        return c, keep

    def geometry(self) -> _abi.Geometry:
        """What SyllableDetector.init derives and checks (SyllableDetector.swift:42-60)."""
        c, keep = self.to_abi()
        g = _abi.Geometry()
        check(_abi.lib.syldet_config_geometry(C.byref(c), C.byref(g)))
        del keep
        return g


def frequencyIndexRange(fourierLength: int, samplingRate: float, startFreq: float, endFreq: float):
    """CircularShortTimeFourierTransform.frequencyIndexRange, :166-191 -> (f0, f1) or None."""
    f0, f1 = C.c_int32(), C.c_int32()
    r = check(_abi.lib.syldet_frequency_index_range(fourierLength, samplingRate, startFreq, endFreq,
                                                    C.byref(f0), C.byref(f1)))
    return (f0.value, f1.value) if r == 1 else None


def createWindow(window: int, length: int) -> np.ndarray:
    """WindowType.createWindow, CircularShortTimeFourierTransform.swift:19-28."""
    out = np.zeros(length, np.float32)
    check(_abi.lib.syldet_make_window(window, length, out.ctypes.data_as(_abi.c_float_p)))
    return out
