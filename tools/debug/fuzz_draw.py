"""Diagnostic: re-run one draw of tests/test_fuzz_gpu.py::test_random_configuration and print where the error sits.
usage: python tools/debug/fuzz_draw.py SEED [classic]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import pyoracle as po, util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth
import test_fuzz_gpu as t

seed = int(sys.argv[1])
rng = np.random.default_rng(9000 + seed)
cfg = t.draw(rng)
C = int(rng.integers(1, 4))
hop = max(0, -cfg.windowOverlap) + cfg.windowLength - max(0, cfg.windowOverlap)
frames = int(rng.integers(cfg.timeRange, 700))
S = max(0, -cfg.windowOverlap) + cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
level = float(10.0 ** rng.uniform(-3, 1))
x = synth.channels(C, S, first=seed * 7, fs=t.FS) * level
stepped = rng.random() < 0.4
env = np.ones(S)
if stepped:
    for _ in range(int(rng.integers(1, 4))):
        at = int(rng.integers(0, S))
        env[at:] *= float(10.0 ** rng.uniform(-2.5, 2.5))
    x = x * np.clip(env, 1e-3, 1e3)[None, :]
x = x.astype(np.float32)
print("W", cfg.windowLength, "hop", hop, "T", cfg.timeRange, "frames", frames, "C", C, "level", level, "stepped", stepped,
      "env", sorted(set(np.clip(env, 1e-3, 1e3).tolist())))
o = util.oracle_for(cfg)
with sd.SyllableDetector(cfg, channels=C) as det:
    det.profile(True)
    out, fl = det.run(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    print(det.lastTimings(), det.fixupStats())
    out = out.cpu().numpy()
for c in range(C):
    w32, _, w64 = o.run(x[c], po.F64, cfg.rule)
    err = np.abs(out[c] - w64) / np.maximum(1.0, np.abs(w64))
    own = np.abs(w32 - w64) / np.maximum(1.0, np.abs(w64))
    e = err.max(axis=1)
    bad = np.nonzero(e > 1e-5)[0]
    print("channel", c, "max err", e.max(), "own", own.max(), "bad evals", bad[:20], len(bad))
    cols64 = o.spectrogram(x[c], po.F64)
    cols32 = o.spectrogram(x[c], po.F32)
    with sd.SyllableDetector(cfg, channels=C) as det2:
        colg = det2.spectrogram(torch.from_numpy(x).cuda()).cpu().numpy()[c]
    for b in bad[:3]:
        T = cfg.timeRange
        print("   window column norms (anchor)", np.sqrt((cols64[b:b + T] ** 2).sum(1)))
        print("   column err fp32 port", np.abs(cols32[b:b + T] - cols64[b:b + T]).max(1))
        print("   column err gpu spect ", np.abs(colg[b:b + T] - cols64[b:b + T]).max(1))
        print("   pass-level: frames", b, "..", b + T - 1, "pass", b // 128, "sample max per frame", [float(np.abs(x[c][(b + t) * hop:(b + t) * hop + cfg.windowLength]).max()) for t in range(T)])
        lo = (b // 128) * 128
        print("   max |x| in pass", float(np.abs(x[c][lo * hop:(lo + 128) * hop + cfg.windowLength]).max()))
    for b in bad[:5]:
        print("   e", b, "out", out[c][b], "w64", w64[b], "w32", w32[b], "env at", env[(b + cfg.timeRange) * hop])
        # where the error is made: the GPU's columns through the anchor's network (fp64), the fp32 port's columns through it,
        # and the anchor's columns through the fp32 port's network
        T = cfg.timeRange
        sc = (lambda v: np.log(v) if cfg.spectrogramScaling == "log" else 20 * np.log10(v) if cfg.spectrogramScaling == "db" else v)
        for label, cols, prec in (("gpu columns, fp64 network", colg, po.F64), ("port columns, fp64 network", cols32, po.F64), ("anchor columns, fp32 network", cols64, po.F32),
                                  ("gpu columns, fp32 network", colg, po.F32)):
            y = o.net_apply(sc(cols[b:b + T].reshape(-1).astype(np.float64)).astype(np.float32 if prec == po.F32 else np.float64), prec)
            print("      %-30s err %.3g" % (label, float((np.abs(y - w64[b]) / np.maximum(1.0, np.abs(w64[b]))).max())))
    print("   network:", [(L.inputs, L.outputs, L.transferFunction) for L in cfg.net.layers], [f.function for f in cfg.net.inputProcessing], "window", cfg.window, "N", cfg.fourierLength)
