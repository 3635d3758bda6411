#!/usr/bin/env python3
"""Diagnostic: the kernel time of the benchmark batch launch by launch from a cold start (the clock governor's ramp, the boost
window, the sustained state): per-launch HIP-event times, printed as means over stretches.
    python tools/ramp_probe.py [launches] [sample|config3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = sys.argv[2] if len(sys.argv) > 2 else "sample"
cfg, C, S = (nets.config3(), 512, 1 << 21) if wl == "config3" else (nets.from_npz(), 64, 1 << 24)
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C) as det:
    E = det.countEvaluations(S)
    out = torch.empty((C, E, det.geometry.outputs), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    time.sleep(2.0)                                   # idle: the clocks fall back
    det.profile(True, history=n)
    t0 = time.perf_counter()
    for _ in range(n):
        det.run(x, out, fl)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [sum(t for _, t in det.timingsOf(back)) for back in range(n)][::-1]
print("%s: %d launches back to back after 2 s idle, wall %.1f ms" % (wl, n, wall * 1e3))
edges = [0, 5, 10, 25, 50, 100, 200, 400, 800, 1600, 3200, 6400, 12800]
t = 0.0
for a, b in zip(edges, edges[1:]):
    if a >= n:
        break
    seg = ms[a:min(b, n)]
    print("launches %5d..%5d  (from t = %7.1f ms)  mean %.4f ms  min %.4f  max %.4f" % (a, min(b, n) - 1, t, sum(seg) / len(seg), min(seg), max(seg)))
    t += sum(seg)
