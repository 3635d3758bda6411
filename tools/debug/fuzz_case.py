"""Diagnostic: one draw of tests/test_fuzz_gpu.py::test_random_configuration in detail.  usage: fuzz_case.py seed [seed ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import pyoracle as po, util
import test_fuzz_gpu as t
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(9000 + seed)
    cfg = t.draw(rng)
    C = int(rng.integers(1, 4))
    hop = max(0, -cfg.windowOverlap) + cfg.windowLength - max(0, cfg.windowOverlap)
    frames = int(rng.integers(cfg.timeRange, 700))
    S = max(0, -cfg.windowOverlap) + cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
    level = float(10.0 ** rng.uniform(-3, 1))
    x = synth.channels(C, S, first=seed * 7, fs=t.FS) * level
    stepped = rng.random() < 0.4
    if stepped:
        env = np.ones(S)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, S))
            env[at:] *= float(10.0 ** rng.uniform(-2.5, 2.5))
        x = x * np.clip(env, 1e-3, 1e3)[None, :]
    x = x.astype(np.float32)
    print("seed", seed, "N W ov hop", cfg.fourierLength, cfg.windowLength, cfg.windowOverlap, hop, "T", cfg.timeRange, cfg.spectrogramScaling, "window", cfg.window, "spectrum", cfg.spectrum,
          "chain", [f.function for f in cfg.net.inputProcessing], [(l.inputs, l.outputs, l.transferFunction) for l in cfg.net.layers], [f.function for f in cfg.net.outputProcessing],
          "C", C, "frames", frames, "level %.3g" % level, "stepped", stepped)
    o = util.oracle_for(cfg)
    with sd.SyllableDetector(cfg, channels=C) as det:
        det.profile(True)
        out, fl = det.run(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        print("  engine", det.geometry.engine, det.lastTimings(), "fixups", det.fixupStats())
        out = out.cpu().numpy()
    for c in range(C):
        _, _, w64 = o.run(x[c], po.F64, cfg.rule)
        o32 = o.run(x[c], po.F32, cfg.rule)[0]
        cols = o.spectrogram(x[c], po.F64)
        ok = np.isfinite(w64).all(axis=1)
        err = np.zeros(len(w64)); own = np.zeros(len(w64))
        err[ok] = (np.abs(out[c][ok] - w64[ok]) / np.maximum(1, np.abs(w64[ok]))).max(axis=1)
        own[ok] = (np.abs(o32[ok] - w64[ok]) / np.maximum(1, np.abs(w64[ok]))).max(axis=1)
        e = int(np.argmax(err))
        T = cfg.timeRange
        print("  ch", c, "worst eval", e, "err %.3g own %.3g" % (err[e], own[e]), "out", w64[e], "col max in window %.3g, col min-of-frame-max %.3g" % (cols[e:e + T].max(), cols[e:e + T].max(axis=1).min()),
              "median err %.3g median own %.3g" % (np.median(err[ok]), np.median(own[ok])), "count err>1e-5:", int((err > 1e-5).sum()), "own>5e-6:", int((own > 5e-6).sum()))
