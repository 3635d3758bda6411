#!/usr/bin/env python3
"""Debug: what the exact (fp64) recomputation path costs a recording it really applies to -- a network WITHOUT a normaliser in front
(the fold kernel hands the windows whose level puts 2^-21 of it past 1e-5 to the fix-up kernel) at several levels:
    python tools/debug/fixup_cost.py [channels] [log2 samples]
Prints, per level: work items of the last run, the share of evaluations they are, milliseconds per run (events around five runs), the kernels syldet_profile lists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 22)
base = nets.from_npz()
rng = np.random.default_rng(3)
for chain in ((), ("mapminmax",), ("l2normalize",)):
    cfg = nets.variant(base, net=nets.random_net(rng, 290, (4,), 1, in_fns=chain))
    x1 = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
    for level in (1e-3, 0.1, 1.0, 10.0):
        x = x1 * level
        with sd.SyllableDetector(cfg, channels=C) as det:
            det.profile(True)
            for _ in range(3):
                out, fl = det.run(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out, fl = det.run(x)
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / 5
            items, over = det.fixupStats()
            E = out.shape[1] * C
            t = det.lastTimings()
        print("chain %-16s level %-6g items %8d (%.3f %% of %d evaluations, 16 a work item) overflow %d   %.3f ms a run;  %s" %
              (",".join(chain) or "(none)", level, items, 100.0 * items * 16 / E, E, over, wall, "  ".join("%s %.3f ms" % (n, ms) for n, ms in t)), flush=True)
