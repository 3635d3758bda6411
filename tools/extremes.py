"""Diagnostic (run on the GPU box): both fused kernels against the oracle on extreme inputs -- zeros, NaN, inf, 1e+-30, level steps of 1e+-12."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'oracle')); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import pyoracle as po, util
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth
cfg = util.sample_net()
S = 64 * 132 * 6 + 500
base = synth.syllable_channel(S, util.template(), seed=5).astype(np.float32)
cases = {"plain": base.copy()}
z = base.copy(); z[20000:30000] = 0.0; cases["zero stretch"] = z
z = base.copy(); z[:] = 0.0; cases["all zero"] = z
z = base.copy(); z[25000] = np.nan; cases["one NaN"] = z
z = base.copy(); z[25000] = np.inf; cases["one inf"] = z
cases["x 1e30"] = (base * np.float32(1e30)).astype(np.float32)
cases["x 1e-30"] = (base * np.float32(1e-30)).astype(np.float32)
z = base.copy(); z[S // 2:] *= np.float32(1e-12); cases["step 1e-12"] = z
z = base.copy(); z[S // 2:] *= np.float32(1e12); cases["step 1e12"] = z
o = util.oracle_for(cfg)
for kern_env in (None, "1"):
    if kern_env: os.environ["SYLDET_FUSED_CLASSIC"] = "1"
    for name, x in cases.items():
        with sd.SyllableDetector(cfg, channels=1) as det:
            det.profile(True)
            out, fl = det.run(torch.from_numpy(x[None]).cuda()); torch.cuda.synchronize()
            k = det.lastTimings()[0][0]
        out = out.cpu().numpy()[0]; fl = fl.cpu().numpy()[0]
        w32, wfl, w64 = o.run(x, po.F64)
        ok = np.isfinite(w64).all(axis=1); okg = np.isfinite(out).all(axis=1)
        both = ok & okg
        err = float(np.abs(out[both] - w64[both]).max()) if both.any() else 0.0
        print("%-14s %-14s finite oracle %5d gpu %5d mismatch %4d  max err %.2e  flags differ %d" % (k, name, ok.sum(), okg.sum(), int((ok != okg).sum()), err, int((fl[both] != wfl[both]).sum())))
