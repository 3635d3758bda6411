"""Diagnostic: one launch of a workload, to be run under `rocprofv3 --kernel-trace --stats` (shows the instantiation that ran)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth
ov = int(sys.argv[1]) if len(sys.argv) > 1 else 124
cfg = nets.variant(nets.from_npz(), windowOverlap=ov)
x = synth.channels_on_device(8, 1 << 20, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=8) as det:
    for _ in range(3):
        det.run(x)
    torch.cuda.synchronize()
