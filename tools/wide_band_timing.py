#!/usr/bin/env python3
"""Diagnostic: the benchmark batch under a band of 58 bins ((1000, 11000) Hz at 256-point frames: more than the 32 bins of
the once-folded kernels): the fold kernel's two-row-tile form against what ran such bands before (SYLDET_FUSED_NOFOLD2=1: the
generic engine's two launches), one box, interleaved.    python tools/wide_band_timing.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth
from syllable_detector_swift_amd.config import frequencyIndexRange
base = nets.from_npz()
lo, hi = 1000.0, 11000.0
f0, f1 = frequencyIndexRange(256, 44100.0, lo, hi)
F = f1 - f0
cfg = nets.variant(base, freqRange=(lo, hi), net=nets.random_net(np.random.default_rng(1), F * 10, (4,), 1))
C, S = 64, 1 << 24
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C) as det:
    E = det.countEvaluations(S); J = det.countFrames(S)
    out = torch.empty((C, E, 1), dtype=torch.float32, device="cuda"); fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    det.profile(True, history=40)
    for i in range(60): det.run(x, out, fl)
    torch.cuda.synchronize()
    tot = {}
    for back in range(40):
        for nm, ms in det.timingsOf(back): tot.setdefault(nm, []).append(ms)
    ms = sum(sum(v) / len(v) for v in tot.values())
    print("bins %%d  %%s  %%.3f ms a step  %%.3g frames/s  %%.3f of 8 TB/s" %% (F, " + ".join("%%s %%.3f" %% (k, sum(v) / len(v)) for k, v in tot.items()), ms, C * J / ms * 1e3, C * J * 533 / (ms * 1e-3) / 8e12))
''' % ROOT
for rnd in range(2):
    for name, env in (("two row tiles", {}), ("two launches", {"SYLDET_FUSED_NOFOLD2": "1"})):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("round %d  %-14s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
