#!/usr/bin/env python3
"""Regenerates tests/golden/* (run in the build container only; needs /root/reference -- except `make_golden.py deep`,
which adds the deeper networks' cases from the committed sample_net.npz).

What it writes
  sample_net.npz          the reference's example network (sample.txt) re-encoded as arrays
                          (weights/offsets/gains/threshold are data; no source text is copied);
  syllable_template.npy   a [10][29] spectrogram pattern the sample network responds to, found by
                          maximising its output (used to synthesise audio that raises detections);
  case_*.npz              per case: configuration, input description (seed / explicit samples +
                          sha256), and the oracle's results (fp64 anchor): first spectrogram
                          columns, network outputs, flags, detection indices.

The outputs come from oracle/ (our restatement), cross-checked in tests/test_oracle.py against an
independent numpy implementation and analytic known answers.  PARITY UNPINNED: the reference
ships no golden vectors and cannot run here.
"""
import hashlib
import os
import sys

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pyoracle as po                                   # noqa: E402
from syllable_detector_swift_amd import nets, synth     # noqa: E402
from syllable_detector_swift_amd.config import NeuralNet, SyllableDetectorConfig  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF_SAMPLE = "/root/reference/sample.txt"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_template(net) -> np.ndarray:
    f = net["inputs"][1]
    xo, g = f["xOffsets"].astype(np.float64), f["gains"].astype(np.float64)
    L0, L1 = net["layers"]
    W0, b0 = L0["weights"].astype(np.float64).reshape(L0["outputs"], L0["inputs"]), L0["biases"].astype(np.float64)
    W1, b1 = L1["weights"].astype(np.float64).reshape(1, -1), L1["biases"].astype(np.float64)

    def fwd(v):
        u = v / np.sqrt((v * v).sum())
        h = np.tanh(W0 @ ((u - xo) * g - 1.0) + b0)
        return ((W1 @ h + b1)[0] + 1.0) / 2.0
    rng = np.random.default_rng(0)
    best = None
    for _ in range(4):
        r = minimize(lambda p: -fwd(np.exp(p)), rng.standard_normal(290) * 0.5, method="L-BFGS-B", options={"maxiter": 500})
        if best is None or r.fun < best.fun:
            best = r
    v = np.exp(best.x)
    return (v / np.linalg.norm(v)).reshape(10, 29).astype(np.float32)


def write_case(name, cfg: SyllableDetectorConfig, samples: np.ndarray, explicit: bool, desc: dict, rule=po.RULE_FIRST):
    o = po.Oracle(po.from_config(cfg))
    out32, fl, o64 = o.run(samples, po.F64, rule)
    cols = o.spectrogram(samples, po.F64)
    thr = np.asarray(cfg.thresholds, np.float64)
    margin = float(np.min(np.abs(o64 - thr[None, :]))) if o64.size else float("inf")
    tmp = os.path.join(GOLD, "_cfg_tmp.npz")
    nets.to_npz(cfg, tmp)
    cfgz = dict(np.load(tmp))
    os.remove(tmp)
    d = {"cfg_" + k: v for k, v in cfgz.items()}
    d.update(window=np.array([cfg.window], np.int32), spectrum=np.array([cfg.spectrum], np.int32),
             rule=np.array([rule], np.int32), n_samples=np.array([samples.size], np.int64),
             sha256=np.array(sha(samples)), columns_head=cols[:4], outputs64=o64[:4096],
             outputs64_stride=o64[::max(1, len(o64) // 512)], flags=np.packbits(fl), n_evals=np.array([len(fl)], np.int64),
             det_0=o.detections(fl, 0.0), det_50ms=o.detections(fl, 0.05), margin=np.array([margin]),
             **{"desc_" + k: np.array(v) for k, v in desc.items()})
    if explicit:
        d["samples"] = samples
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **d)
    print("%-28s S=%-8d J=%-6d E=%-6d detections=%-4d margin=%.3g" %
          (name, samples.size, o.count_frames(samples.size), len(fl), int(fl.sum()), margin))


def main():
    os.makedirs(GOLD, exist_ok=True)
    net = po.parse_text(open(REF_SAMPLE).read())
    # sample.txt -> arrays (through the product dataclass only as a container)
    from syllable_detector_swift_amd.config import NeuralNetLayer, ProcessingFunction

    def fn(f):
        return ProcessingFunction(f["function"], f.get("xOffsets"), f.get("gains"), float(f.get("y", 0.0)))
    cfg = SyllableDetectorConfig(net["samplingRate"], net["fourierLength"], net["windowLength"], net["windowOverlap"],
                                 net["freqRange"], net["timeRange"], net["scaling"], [float(t) for t in net["thresholds"]],
                                 NeuralNet([NeuralNetLayer(L["inputs"], L["outputs"], L["weights"].reshape(L["outputs"], L["inputs"]),
                                                           L["biases"], L["transferFunction"]) for L in net["layers"]],
                                           [fn(f) for f in net["inputs"]], [fn(f) for f in net["outputs"]]))
    nets.to_npz(cfg, os.path.join(GOLD, "sample_net.npz"))
    tpl_path = os.path.join(GOLD, "syllable_template.npy")
    template = make_template(net)
    np.save(tpl_path, template)

    base = nets.from_npz(os.path.join(GOLD, "sample_net.npz"))
    rng = np.random.default_rng(2024)

    # A: the sample network on explicit syllable audio (2 s)
    xa = synth.syllable_channel(88200, template, seed=5)
    write_case("case_sample_syllables", base, xa, True, {"kind": "syllable_channel", "seed": 5})
    # B: BASELINE audio (FM bursts), channel 0, seeded
    xb = synth.channel(1 << 18, 0)
    write_case("case_sample_fmburst", base, xb, False, {"kind": "channel", "seed": 0})
    # C: hop-128 variant of the benchmark line
    xc = synth.syllable_channel(1 << 17, template, seed=6, hop=128)
    write_case("case_sample_hop128", nets.variant(base, windowOverlap=128), xc, False,
               {"kind": "syllable_channel_hop128", "seed": 6})
    # D: BASELINE config 3 (1024-pt, hop 256)
    xd = synth.channel(1 << 17, 3)
    write_case("case_config3", nets.config3(), xd, False, {"kind": "channel", "seed": 3})
    # E: processing-chain / option variants on a small front-end (N=128, W=96 zero-padded, gap)
    small = dict(samplingRate=16000.0, fourierLength=128, windowLength=96, freqRange=(500.0, 4000.0), timeRange=6)
    f0, f1 = 4, 33                         # ceil(128/16000*500)=4, floor(128/16000*4000)+1=33
    I = (f1 - f0) * 6
    xe = synth.channel(1 << 15, 9, fs=16000.0)
    variants = {
        "case_chain_normalize_db": dict(windowOverlap=32, scaling="db", in_fns=("normalize",), tf=("LogSig", "PureLin"), outs=1),
        "case_chain_normstd_log": dict(windowOverlap=-16, scaling="log", in_fns=("normalizestd", "mapstd"), tf=("SatLin", "TanSig"), outs=3),
        "case_chain_mapstd_only": dict(windowOverlap=80, scaling="linear", in_fns=("mapstd",), tf=("TanSig", "LogSig"), outs=2),
        "case_chain_none": dict(windowOverlap=0, scaling="linear", in_fns=(), tf=("TanSig", "PureLin"), outs=1),
    }
    for name, v in variants.items():
        nn = nets.random_net(rng, I, (5,), v["outs"], transfer=v["tf"], in_fns=v["in_fns"],
                             out_fns=("mapminmax",) if v["outs"] == 1 else ("mapstd", "mapminmax"))
        c = SyllableDetectorConfig(small["samplingRate"], small["fourierLength"], small["windowLength"], v["windowOverlap"],
                                   small["freqRange"], small["timeRange"], v["scaling"], [0.4] * v["outs"], nn)
        write_case(name, c, xe, False, {"kind": "channel_fs16000", "seed": 9},
                   rule=po.RULE_ANY if v["outs"] > 1 else po.RULE_FIRST)


def deep_cases():
    """Networks of three and four layers (NeuralNet.apply runs any layerCount: NeuralNet.swift:310-313,
    SyllableDetectorConfig.swift:232-259).  Made from the committed sample_net.npz: needs no reference tree
    (python tests/golden/make_golden.py deep)."""
    base = nets.from_npz(os.path.join(GOLD, "sample_net.npz"))
    template = np.load(os.path.join(GOLD, "syllable_template.npy"))
    rng = np.random.default_rng(515)
    # three layers behind the example front-end, on audio that holds syllables; first-output rule
    n3 = nets.random_net(rng, 290, (6, 3), 1, transfer=("TanSig", "LogSig", "PureLin"), in_fns=("l2normalize", "mapminmax"), out_fns=("mapminmax",))
    xa = synth.syllable_channel(1 << 17, template, seed=7)
    write_case("case_deep3_sample_front", nets.variant(base, net=n3, thresholds=[0.416194]), xa, False, {"kind": "syllable_channel", "seed": 7})
    # four layers, two outputs under the any-output rule, log columns behind normalizestd, the small front-end
    small = dict(samplingRate=16000.0, fourierLength=128, windowLength=96, freqRange=(500.0, 4000.0), timeRange=6)
    I = (33 - 4) * 6
    n4 = nets.random_net(rng, I, (8, 4, 3), 2, transfer=("SatLin", "TanSig", "LogSig", "PureLin"), in_fns=("normalizestd", "mapstd"),
                         out_fns=("mapstd", "mapminmax"))
    c4 = SyllableDetectorConfig(small["samplingRate"], small["fourierLength"], small["windowLength"], 32, small["freqRange"], small["timeRange"],
                                "log", [0.958960, 0.359288], n4)
    xe = synth.channel(1 << 15, 9, fs=16000.0)
    write_case("case_deep4_small_front_any", c4, xe, False, {"kind": "channel_fs16000", "seed": 9}, rule=po.RULE_ANY)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deep":
        deep_cases()
    else:
        main()
        deep_cases()
