#!/usr/bin/env python3
"""Diagnostic: energy per wave-instruction on this MI355X.  Runs tools/ubench/energy (one loop kind on every CU, random
operands, back to back for a few seconds) per mode, samples socket power and shader clock with rocm-smi from a side thread,
and prints watts over the `idle` mode's (every CU occupied by sleeping waves) divided by the loop's instruction rate.

    hipcc --offload-arch=gfx950 -O3 -o tools/ubench/energy tools/ubench/energy.hip
    python tools/energy_probe.py [seconds] [mode ...]
"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
modes = sys.argv[2:] or ["idle", "mfma16", "mfma16w2", "mfma32", "mfma32w2", "fma", "split", "ldsr", "m16fma", "hbm", "hbmnt"]
exe = os.path.join(ROOT, "tools", "ubench", "energy")


def probe(mode):
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
            p = re.search(r"Power \(W\): ([0-9.]+)", r)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r)
            samples.append((time.time(), float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))

    th = threading.Thread(target=sampler)
    t0 = time.time()
    th.start()
    out = subprocess.run([exe, mode, str(secs)], capture_output=True, text=True).stdout.strip()
    t1 = time.time()
    stop[0] = True
    th.join()
    busy = [(p, c) for t, p, c in samples if t0 + 1.5 <= t <= t1 - 0.3 and p > 0]
    watts = sum(p for p, _ in busy) / max(len(busy), 1)
    sclk = sum(c for _, c in busy) / max(len(busy), 1)
    m = re.search(r"([0-9.e+]+) wave-instructions/s", out)
    rate = float(m.group(1)) if m else 0.0
    return out, watts, sclk, rate, len(busy)


base = None
for mode in modes:
    out, watts, sclk, rate, n = probe(mode)
    print(out)
    if mode == "idle":
        base = watts
    line = "    power %.0f W, sclk %.0f MHz (%d samples)" % (watts, sclk, n)
    if base is not None and mode != "idle" and rate > 0:
        line += "; over idle %.0f W = %.2f nJ per wave-instruction" % (watts - base, (watts - base) / rate * 1e9)
    print(line, flush=True)
    time.sleep(1.0)
