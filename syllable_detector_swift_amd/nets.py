"""Builders for the network configurations BASELINE.md names.

- `from_npz`: the reference's example detector (sample.txt: 44.1 kHz, N=W=256, overlap 124,
  2-7 kHz, T=10, linear, l2normalize -> mapminmax, 290 -> 4 TanSig -> 1 PureLin, output
  mapminmax) re-encoded as arrays in tests/golden/sample_net.npz by tests/golden/make_golden.py;
- `config3`: synthetic N=W=1024, overlap 768 (hop 256), 1160 -> 4 -> 1 (BASELINE config 3);
- `wide_mlp`: sample front-end with a 290 -> 4096 -> 1 network (BASELINE config 5);
- `variant`: small edits of a configuration for parity cases.
"""
from __future__ import annotations

import copy
import os

import numpy as np

from . import _abi
from .config import NeuralNet, NeuralNetLayer, ProcessingFunction, SyllableDetectorConfig

_FN_NAMES = ["l2normalize", "normalize", "normalizestd", "mapminmax", "mapstd"]
_TF_NAMES = ["TanSig", "LogSig", "PureLin", "SatLin"]
_SCALING = ["linear", "log", "db"]

DEFAULT_SAMPLE_NET = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                  "tests", "golden", "sample_net.npz")


def to_npz(cfg: SyllableDetectorConfig, path: str) -> None:
    d = {"scalars": np.array([cfg.samplingRate, cfg.fourierLength, cfg.windowLength, cfg.windowOverlap,
                              cfg.freqRange[0], cfg.freqRange[1], cfg.timeRange,
                              _SCALING.index(cfg.spectrogramScaling)], np.float64),
         "thresholds": np.asarray(cfg.thresholds, np.float64),
         "n": np.array([len(cfg.net.inputProcessing), len(cfg.net.layers), len(cfg.net.outputProcessing)], np.int32)}
    for tag, fl in (("in", cfg.net.inputProcessing), ("out", cfg.net.outputProcessing)):
        for i, f in enumerate(fl):
            d["%s%d_kind" % (tag, i)] = np.array([_FN_NAMES.index(f.function)], np.int32)
            if f.function in ("mapminmax", "mapstd"):
                d["%s%d_xoff" % (tag, i)] = np.asarray(f.xOffsets, np.float32)
                d["%s%d_gain" % (tag, i)] = np.asarray(f.gains, np.float32)
                d["%s%d_y" % (tag, i)] = np.array([f.y], np.float32)
    for i, L in enumerate(cfg.net.layers):
        d["layer%d_w" % i] = np.asarray(L.weights, np.float32).reshape(L.outputs, L.inputs)
        d["layer%d_b" % i] = np.asarray(L.biases, np.float32)
        d["layer%d_tf" % i] = np.array([_TF_NAMES.index(L.transferFunction)], np.int32)
    np.savez(path, **d)


def from_npz(path: str = DEFAULT_SAMPLE_NET) -> SyllableDetectorConfig:
    z = np.load(path)
    s = z["scalars"]
    n_in, n_layers, n_out = [int(v) for v in z["n"]]

    def fns(tag, n):
        out = []
        for i in range(n):
            name = _FN_NAMES[int(z["%s%d_kind" % (tag, i)][0])]
            if name in ("mapminmax", "mapstd"):
                out.append(ProcessingFunction(name, z["%s%d_xoff" % (tag, i)], z["%s%d_gain" % (tag, i)],
                                              float(z["%s%d_y" % (tag, i)][0])))
            else:
                out.append(ProcessingFunction(name))
        return out

    layers = []
    for i in range(n_layers):
        w = z["layer%d_w" % i]
        layers.append(NeuralNetLayer(int(w.shape[1]), int(w.shape[0]), w, z["layer%d_b" % i],
                                     _TF_NAMES[int(z["layer%d_tf" % i][0])]))
    return SyllableDetectorConfig(float(s[0]), int(s[1]), int(s[2]), int(s[3]), (float(s[4]), float(s[5])), int(s[6]),
                                  _SCALING[int(s[7])], [float(t) for t in z["thresholds"]],
                                  NeuralNet(layers, fns("in", n_in), fns("out", n_out)))


def _dense(rng, outputs, inputs):
    return (rng.standard_normal((outputs, inputs)) / np.sqrt(inputs)).astype(np.float32)


def config3(seed: int = 7) -> SyllableDetectorConfig:
    """BASELINE config 3: 1024-pt FFT, hop 256, bins [47,163), 1160 -> 4 TanSig -> 1 PureLin."""
    rng = np.random.default_rng(seed)
    F, T = 116, 10
    I = F * T
    layers = [NeuralNetLayer(I, 4, _dense(rng, 4, I), (0.1 * rng.standard_normal(4)).astype(np.float32), "TanSig"),
              NeuralNetLayer(4, 1, _dense(rng, 1, 4), (0.1 * rng.standard_normal(1)).astype(np.float32), "PureLin")]
    inp = [ProcessingFunction("l2normalize"),
           ProcessingFunction("mapminmax", np.zeros(I, np.float32), np.full(I, 2.0, np.float32), -1.0)]
    return SyllableDetectorConfig(44100.0, 1024, 1024, 768, (2000.0, 7000.0), T, "linear", [0.5],
                                  NeuralNet(layers, inp, []))


def wide_mlp(base: SyllableDetectorConfig, hidden: int = 4096, seed: int = 11) -> SyllableDetectorConfig:
    """BASELINE config 5: the base front-end with an I -> hidden TanSig -> 1 PureLin network."""
    rng = np.random.default_rng(seed)
    I = base.net.inputs
    cfg = copy.deepcopy(base)
    cfg.net = NeuralNet([NeuralNetLayer(I, hidden, _dense(rng, hidden, I), np.zeros(hidden, np.float32), "TanSig"),
                         NeuralNetLayer(hidden, 1, _dense(rng, 1, hidden), np.zeros(1, np.float32), "PureLin")],
                        copy.deepcopy(base.net.inputProcessing), [])
    cfg.thresholds = [0.5]
    return cfg


def variant(base: SyllableDetectorConfig, **changes) -> SyllableDetectorConfig:
    cfg = copy.deepcopy(base)
    for k, v in changes.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def random_net(rng, inputs: int, hidden, outputs: int, transfer=("TanSig", "PureLin"), in_fns=("l2normalize", "mapminmax"),
               out_fns=("mapminmax",)) -> NeuralNet:
    """A random network of the given shape with the given processing chain (parity cases)."""
    sizes = [inputs] + list(hidden) + [outputs]
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(NeuralNetLayer(sizes[i], sizes[i + 1], _dense(rng, sizes[i + 1], sizes[i]),
                                     (0.2 * rng.standard_normal(sizes[i + 1])).astype(np.float32),
                                     transfer[min(i, len(transfer) - 1)] if i < len(sizes) - 2 else transfer[-1]))

    def fn(name, n):
        if name in ("mapminmax", "mapstd"):
            return ProcessingFunction(name, (0.01 * rng.standard_normal(n)).astype(np.float32),
                                      (1.0 + rng.random(n)).astype(np.float32), float(-1.0 if name == "mapminmax" else 0.25))
        return ProcessingFunction(name)
    return NeuralNet(layers, [fn(f, inputs) for f in in_fns], [fn(f, outputs) for f in out_fns])
