import sys, os
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import numpy as np, torch
import pyoracle as po, util
import syllable_detector_swift_amd as sd
cfg, x, gold = util.load_case("case_sample_syllables")
x = x[:120000]
o = util.oracle_for(cfg)
_, wfl, w64 = o.run(x, po.F64)
cols64 = o.spectrogram(x, po.F64)
with sd.SyllableDetector(cfg, channels=1) as det:
    det.profile(True)
    out, fl = det.run(torch.from_numpy(x[None,:]).cuda())
    torch.cuda.synchronize()
    print(det.lastTimings())
    out = out.cpu().numpy()[0]; fl = fl.cpu().numpy()[0]
err = np.abs(out - w64).max()
print("max err", err, "flags equal", (fl == wfl).all(), "detections", int(fl.sum()), int(wfl.sum()))
