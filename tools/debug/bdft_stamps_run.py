import os, sys, torch
sys.path.insert(0, os.getcwd())
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth
cfg, C, S = nets.config3(), 512, 1 << 21
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C) as det:
    for _ in range(3):
        det.run(x)
    torch.cuda.synchronize()
