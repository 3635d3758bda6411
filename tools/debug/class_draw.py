"""Diagnostic: re-run one draw of tests/test_fuzz_gpu.py::test_random_example_class_detector_on_the_register_resident_kernel.
usage: python tools/debug/class_draw.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import pyoracle as po, util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth
import test_fuzz_gpu as t

seed = int(sys.argv[1])
rng = np.random.default_rng(77000 + seed)
cfg = t.draw_example_class(rng)
hop = cfg.windowLength - cfg.windowOverlap
edges = [10, 11, 63, 64, 65, 73, 74, 127, 128, 129, 137, 192, 201, 2047, 2048, 2049, 2057, 2058, 4100]
frames = max(cfg.timeRange, int(edges[seed % len(edges)] if seed < len(edges) else rng.integers(10, 6000)))
S = cfg.windowLength + (frames - 1) * hop + int(rng.integers(0, hop))
C = int(rng.integers(1, 4))
x = synth.channels(C, S, first=seed * 5, fs=t.FS) * float(10.0 ** rng.uniform(-3, 1))
env = np.ones(S)
if rng.random() < 0.6:
    for _ in range(int(rng.integers(1, 5))):
        env[int(rng.integers(0, S)):] *= float(10.0 ** rng.uniform(-2.5, 2.5))
    x = x * np.clip(env, 1e-3, 1e3)[None, :]
x = x.astype(np.float32)
print("W", cfg.windowLength, "hop", hop, "T", cfg.timeRange, "F", cfg.net.layers[0].inputs // cfg.timeRange, "frames", frames, "C", C,
      "H", cfg.net.layers[0].outputs, [f.function for f in cfg.net.inputProcessing], "env", sorted(set(np.clip(env, 1e-3, 1e3).tolist())))
o = util.oracle_for(cfg)
with sd.SyllableDetector(cfg, channels=C) as det:
    det.profile(True)
    out, fl = det.run(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    print(det.lastTimings(), det.fixupStats())
    out = out.cpu().numpy()
    os.environ["SYLDET_FUSED_CLASSIC"] = "1"
    outc, _ = det.run(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    print("classic:", det.lastTimings(), det.fixupStats())
    outc = outc.cpu().numpy()
for c in range(C):
    w32, _, w64 = o.run(x[c], po.F64, cfg.rule)
    ok = np.isfinite(w64).all(axis=1)
    err = np.abs(out[c] - w64) / np.maximum(1.0, np.abs(w64))
    errc = np.abs(outc[c] - w64) / np.maximum(1.0, np.abs(w64))
    own = np.abs(w32 - w64) / np.maximum(1.0, np.abs(w64))
    e = np.where(ok, err.max(axis=1), 0)
    bad = np.nonzero(e > max(1e-5, 4 * own[ok].max()))[0]
    print("channel", c, "max err", e.max(), "classic", np.where(ok, errc.max(axis=1), 0).max(), "own", own[ok].max(), "bad evals", bad[:20], len(bad))
    cols64 = o.spectrogram(x[c], po.F64)
    T = cfg.timeRange
    for b in bad[:3]:
        print("   e", b, "out", out[c][b], "classic", outc[c][b], "w64", w64[b], "w32", w32[b])
        print("   window column norms (anchor)", np.sqrt((cols64[b:b + T] ** 2).sum(1)))
        print("   sample max per frame", [float(np.abs(x[c][(b + t) * hop:(b + t) * hop + cfg.windowLength]).max()) for t in range(T)])
        lo = (b // 64) * 64
        print("   pass", b // 64, "max |x| in pass", float(np.abs(x[c][lo * hop:(lo + 64) * hop + cfg.windowLength]).max()),
              "next", float(np.abs(x[c][(lo + 64) * hop:(lo + 128) * hop + cfg.windowLength]).max()) if (lo + 64) * hop < S else None)
