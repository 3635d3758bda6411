#!/usr/bin/env python3
"""Diagnostic: socket power and shader clock (rocm-smi, sampled from a side thread) while the benchmark batch's kernel -- or,
for comparison, a plain streaming read of the same 4.3 GB (torch's sum) -- runs back to back for a few seconds.
    python tools/power_probe.py [seconds] [fused|fusedr|read|classic|config3|config5]     (fused: the symmetric-fold kernel; fusedr: fused_r_kernel)"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
if mode == "classic":
    os.environ["SYLDET_FUSED_CLASSIC"] = "1"
if mode == "fusedr":
    os.environ["SYLDET_FUSED_NOFOLD"] = "1"
samples, stop = [], False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Power \(W\): ([0-9.]+)", r)
        c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r)
        samples.append((time.time(), float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))
base = nets.from_npz()
C, S, engine = 64, 1 << 24, _abi.ENGINE_FUSED
if mode == "config3":
    base, C, S, engine = nets.config3(), 512, 1 << 21, 0
if mode == "config5":
    base, engine = nets.wide_mlp(base), 3
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=base.samplingRate)
with sd.SyllableDetector(base, channels=C, engine=engine) as det:
    E = det.countEvaluations(S)
    out = torch.empty((C, E, det.geometry.outputs), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    def step():
        if mode == "read":
            return x.sum()
        det.run(x, out, fl)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    time.sleep(1.0)
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50):
            step()
        torch.cuda.synchronize(); n += 50
    t1 = time.time()
    time.sleep(1.0)
    stop = True; th.join()
ms = (t1 - t0) / n * 1e3
print("%s: launches %d, %.3f ms each (wall, synchronised every 50) = %.2f TB/s of samples" % (mode, n, ms, C * S * 4 / (ms * 1e-3) / 1e12))
busy = [(p, c) for t, p, c in samples if t0 + 1.0 <= t <= t1]
if busy:
    print("sustained (after the first second): power %.0f W, sclk %.0f MHz (%d samples)" % (sum(p for p, _ in busy) / len(busy), sum(c for _, c in busy) / len(busy), len(busy)))
if "-v" in sys.argv:
    for t, p, c in samples:
        print("t=%6.2f s  %s  power %6.1f W  sclk %4d MHz" % (t - t0, "busy" if t0 <= t <= t1 else "idle", p, c))
