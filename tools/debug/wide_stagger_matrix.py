#!/usr/bin/env python3
"""Debug: the wide GEMM's forms run to run AND against the shipped order, bit for bit, at a size that takes many rounds of
workgroups.  Every variant runs in a child process of its own (the library reads its switches at syldet_create):

    python tools/debug/wide_stagger_matrix.py [--runs 12] [--C 64] [--S 8388608] NAME[:VAR=VAL[,VAR=VAL...]] ...

The first variant is the reference: its first output is kept, every run of every variant is compared with it.  Differing
evaluations are saved (gpurun_out/wide_matrix_<name>_run<k>.npz: indices, values, reference values) for tools/debug/wide_blame.py."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi
name = os.environ["WM_NAME"]; runs = int(os.environ["WM_RUNS"]); C = int(os.environ["WM_C"]); S = int(os.environ["WM_S"])
ref_path = os.environ["WM_REF"]; out_dir = os.environ["WM_OUT"]
cfg = nets.wide_mlp(nets.from_npz())
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
ref = np.load(ref_path) if os.path.exists(ref_path) else None
with sd.SyllableDetector(cfg, channels=C, engine=_abi.ENGINE_WIDE_BF16) as det:
    det.profile(True)
    bad = 0; ms = []
    for k in range(runs):
        o, f = det.run(x)
        torch.cuda.synchronize()
        ms.append([t for n, t in det.lastTimings() if n.startswith("wide_gemm")][0])
        o = o.cpu().numpy()
        if ref is None:
            ref = o.copy(); np.save(ref_path, ref)
            print("  [%%s] kernels: %%s" %% (name, [n for n, _ in det.lastTimings()]), flush=True)
        d = np.argwhere(o[..., 0] != ref[..., 0])
        if len(d):
            bad += 1
            blocks = {(int(c), int(e) // 16) for c, e in d}
            print("  [%%s] run %%d: %%d evaluations differ (%%d blocks of 16; %%d of them whole); max |diff| %%.3g; first %%s" %% (
                name, k, len(d), len(blocks), sum(1 for b in blocks if sum(1 for c, e in d if (int(c), int(e) // 16) == b) == 16) if len(d) < 200000 else -1,
                float(np.abs(o - ref).max()), d[:3].tolist()), flush=True)
            if bad <= 3:
                np.savez(os.path.join(out_dir, "wide_matrix_%%s_run%%d.npz" %% (name, k)), idx=d[:4096], val=o[..., 0][tuple(d[:4096].T)], ref=ref[..., 0][tuple(d[:4096].T)], C=C, S=S)
    ms = sorted(ms)
    print("[%%s] %%d of %%d runs differ from the reference; gemm ms min %%.3f median %%.3f" %% (name, bad, runs, ms[0], ms[len(ms) // 2]), flush=True)
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=12)
ap.add_argument("--C", type=int, default=64)
ap.add_argument("--S", type=int, default=1 << 23)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
ref = "/tmp/wide_matrix_ref_%d_%d.npy" % (a.C, a.S)
if os.path.exists(ref):
    os.remove(ref)
rc = 0
for v in a.variants:
    name, _, envs = v.partition(":")
    env = dict(os.environ, WM_NAME=name, WM_RUNS=str(a.runs), WM_C=str(a.C), WM_S=str(a.S), WM_REF=ref, WM_OUT=out)
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition("=")
        env[k] = val.replace("@ROOT", ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, timeout=900)
    rc |= r.returncode
sys.exit(rc)
