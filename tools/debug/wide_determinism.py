#!/usr/bin/env python3
"""Debug: the wide engine run to run, bit for bit, at a size that takes many rounds of workgroups (a race shows here, not at
test sizes: round 4's staggered GEMM was bit-identical on 8 x 200 000 samples and differed in 1-10 % of the evaluations on
64 x 2^23).      python tools/debug/wide_determinism.py [hidden units] [runs]      (CHECK_C, CHECK_S: channels, samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
C = int(os.environ.get("CHECK_C", "64"))
S = int(os.environ.get("CHECK_S", str(1 << 23)))
base = nets.from_npz()
cfg = nets.wide_mlp(base) if H == 4096 else nets.variant(base, net=nets.random_net(np.random.default_rng(3), 290, (H,), 1))
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C, engine=_abi.ENGINE_WIDE_BF16) as det:
    det.profile(True)
    first = None
    bad = 0
    for k in range(runs):
        o, f = det.run(x)
        torch.cuda.synchronize()
        o = o.cpu().numpy().copy()
        if first is None:
            first = o
            print("kernels:", [n for n, _ in det.lastTimings()])
        elif (o != first).any():
            bad += 1
            d = np.argwhere(o != first)
            print("run %d: %d evaluations differ from run 0; max |diff| %.3g; first %s" % (k, len(d), float(np.abs(o - first).max()), d[:4].tolist()))
print("%d of %d runs differ from the first" % (bad, runs - 1))
sys.exit(1 if bad else 0)
