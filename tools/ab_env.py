#!/usr/bin/env python3
"""Diagnostic: kernel time of a benchmark workload under several environments (switches the library reads at syldet_create) on
the SAME box, interleaved, each in a child process of its own.

    python tools/ab_env.py [--workload sample|config3|config5|hop128] [--rounds 3] NAME[:VAR=VAL[,VAR=VAL...]] ...

e.g.   python tools/ab_env.py fold2 fold1:SYLDET_FUSED_NOFOLD2=1
Prints per round and variant: kernel name, min / median / mean ms over 200 launches after 100 untimed ones, and the socket
power (rocm-smi) sampled while they run.
"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, subprocess, threading, time
sys.path.insert(0, %r)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi
wl = os.environ.get("AB_WORKLOAD", "sample")
engine = 0
if wl == "config3":
    cfg, C, S = nets.config3(), 512, 1 << 21
elif wl == "config5":
    cfg, C, S, engine = nets.wide_mlp(nets.from_npz()), 64, 1 << 24, 3
else:
    cfg, C, S = nets.from_npz(), 64, 1 << 24
    if wl == "hop128":
        cfg = nets.variant(cfg, windowOverlap=128)
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
watts = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            for line in o.splitlines()[1:]:
                p = line.split(",")
                if len(p) > 1:
                    watts.append(float(p[1]))
        except Exception:
            pass
        time.sleep(0.05)
with sd.SyllableDetector(cfg, channels=C, engine=engine) as det:
    E = det.countEvaluations(S)
    out = torch.empty((C, E, det.geometry.outputs), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    n_timed = 200 if wl != "config5" else 30
    det.profile(True, history=n_timed)
    for i in range(100 if wl != "config5" else 10):
        det.run(x, out, fl)
    torch.cuda.synchronize()
    th = threading.Thread(target=poll); th.start()
    t0 = time.perf_counter()
    for i in range(n_timed):
        det.run(x, out, fl)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n_timed * 1e3
    stop = True; th.join()
    per = {}
    for back in range(n_timed):
        for nm, ms in det.timingsOf(back):
            per.setdefault(nm, []).append(ms)
    dom = max(per, key=lambda k: sum(per[k]))
    ms = sorted(per[dom])
    w = sorted(watts)
    print("%%s min %%.4f med %%.4f mean %%.4f ms (wall %%.4f)  power med %%s W (%%d samples)" %% (dom, ms[0], ms[len(ms) // 2], sum(ms) / len(ms), wall,
          ("%%.0f" %% w[len(w) // 2]) if w else "n/a", len(w)))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sample")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
for rnd in range(a.rounds):
    for v in a.variants:
        name, _, envs = v.partition(":")
        env = dict(os.environ, AB_WORKLOAD=a.workload)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("round %d  %-14s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
