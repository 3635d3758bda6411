#!/usr/bin/env python3
"""Diagnostic: kernel time of the benchmark batch under several builds of the library on the SAME box, interleaved.

    python tools/ab_kernel.py lib lib_base [lib_x ...]      (directories under syllable_detector_swift_amd/)

Each build runs in its own child process (the library path is read at import), three rounds each, alternating; prints the
per-round averages of the fused kernel's own timer.
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi
base = nets.from_npz()
C, S = 64, 1 << 24
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=base.samplingRate)
with sd.SyllableDetector(base, channels=C, engine=_abi.ENGINE_FUSED) as det:
    E = det.countEvaluations(S)
    out = torch.empty((C, E, det.geometry.outputs), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    det.profile(True)
    ms = []
    stamped = "SYLDET_FUSED_STAMPS" in os.environ      # (a -DSYLDET_R_STAMPS build: its phase table goes to stderr)
    for i in range(4 if stamped else 300):             # (the first 100 launches: the clock governor's ramp, MEASUREMENTS R3.8)
        det.run(x, out, fl)
        if i >= (2 if stamped else 100):
            ms.append(det.lastTimings()[0][1])
    torch.cuda.synchronize()
    ms.sort()
    print("%%s %%.4f %%.4f %%.4f" %% (det.lastTimings()[0][0], ms[0], ms[len(ms) // 2], sum(ms) / len(ms)))
''' % ROOT

libs = sys.argv[1:] or ["lib", "lib_base"]
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, SYLDET_LIB=os.path.join(ROOT, "syllable_detector_swift_amd", lib, "libsyldet.so"))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("round %d  %-12s %s" % (rnd, lib, (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
        if "SYLDET_FUSED_STAMPS" in os.environ:
            print("\n".join(r.stderr.strip().split("\n")[-9:]), flush=True)
