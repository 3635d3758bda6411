#!/usr/bin/env python3
"""Diagnostic: kernel time of BASELINE configs[2] (1024-point frames, hop 256, 512 channels x 2^21 samples) under several builds of
the library on ONE box, interleaved (each build in a child process: the library path is read at import).
    python tools/c3_timing.py lib lib_x ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth
cfg = nets.config3()
C, S = 512, 1 << 21
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C) as det:
    E = det.countEvaluations(S)
    out = torch.empty((C, E, 1), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    det.profile(True)
    ms = []
    for i in range(30):
        det.run(x, out, fl)
        if i >= 8:
            ms.append(sum(t for _, t in det.lastTimings()))
    torch.cuda.synchronize()
    ms.sort()
    print("%%s %%.4f %%.4f" %% ("+".join(n for n, _ in det.lastTimings()), ms[0], ms[len(ms) // 2]))
''' % ROOT
libs = sys.argv[1:] or ["lib"]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, SYLDET_LIB=os.path.join(ROOT, "syllable_detector_swift_amd", lib, "libsyldet.so"))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("round %d  %-14s %s" % (rnd, lib, (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
