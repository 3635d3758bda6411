import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi
cfg = nets.from_npz()
C, S = 64, 1 << 24
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
with sd.SyllableDetector(cfg, channels=C) as det:
    det.profile(True, history=50)
    for i in range(30):
        cols = det.spectrogram(x)
    torch.cuda.synchronize()
    ms = {}
    for i in range(50):
        cols = det.spectrogram(x)
        torch.cuda.synchronize()
        for nm, t in det.lastTimings():
            ms.setdefault(nm, []).append(t)
    for nm, v in ms.items():
        v.sort()
        print(os.environ.get("SYLDET_FUSED_NOFOLD", "fold"), nm, "min %.4f med %.4f" % (v[0], v[len(v) // 2]))
