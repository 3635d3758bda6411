#!/usr/bin/env python3
"""Debug: every benchmark workload N times at full size, bit for bit against its first run (the suite's
test_full_size_runs_are_reproducible does three runs each):   python tools/debug/determinism_all.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi
from syllable_detector_swift_amd.config import frequencyIndexRange

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
base = nets.from_npz()
f0, f1 = frequencyIndexRange(256, base.samplingRate, 1000.0, 11000.0)
work = {
    "sample": (base, 64, 1 << 24, _abi.ENGINE_AUTO),
    "hop128": (nets.variant(base, windowOverlap=128), 64, 1 << 24, _abi.ENGINE_AUTO),
    "H16x4": (nets.variant(base, net=nets.random_net(np.random.default_rng(5), 290, (16,), 4), thresholds=[0.5] * 4), 64, 1 << 24, _abi.ENGINE_AUTO),
    "wide_band": (nets.variant(base, freqRange=(1000.0, 11000.0), net=nets.random_net(np.random.default_rng(11), (f1 - f0) * 10, (4,), 1)), 64, 1 << 24, _abi.ENGINE_AUTO),
    "config3": (nets.config3(), 512, 1 << 21, _abi.ENGINE_AUTO),
    "config5": (nets.wide_mlp(base), 64, 1 << 24, _abi.ENGINE_WIDE_BF16),
}
# round 5: the kernels behind the switches and the fallback engines too -- a packed instruction that loses a product beside another
# wave's matrix instructions (MEASUREMENTS R5.1) shows as run-to-run differences only at sizes like these, and the 8-wave fused
# kernel's normalizestd statistics had 32 of them
nstd = nets.variant(base, net=nets.random_net(np.random.default_rng(21), 290, (4,), 1, in_fns=("normalizestd", "mapstd")))
env_work = {
    "sample, 8-wave fused kernel": (base, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_CLASSIC": "1"}),
    "normalizestd chain, 8-wave fused": (nstd, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_CLASSIC": "1"}),
    "sample, register-resident kernel": (base, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_NOFOLD": "1"}),
    "sample, once-folded fold kernel": (base, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_NOFOLD2": "1"}),
    "sample, generic engine": (base, 64, 1 << 22, _abi.ENGINE_GENERIC, {}),
    "dB columns (FFT across lanes + MFMA network)": (nets.variant(base, spectrogramScaling="db"), 64, 1 << 22, _abi.ENGINE_AUTO, {}),
    "config3, Blackman window (fft1k)": (nets.variant(nets.config3(), window=_abi.WINDOW_BLACKMAN), 256, 1 << 21, _abi.ENGINE_AUTO, {}),
    "config3, generic engine": (nets.config3(), 128, 1 << 21, _abi.ENGINE_GENERIC, {}),
    "config5, unstaggered": (nets.wide_mlp(base), 64, 1 << 23, _abi.ENGINE_WIDE_BF16, {"SYLDET_WIDE_NOSTAGGER": "1"}),
    "config5, 16-wave workgroups": (nets.wide_mlp(base), 64, 1 << 23, _abi.ENGINE_WIDE_BF16, {"SYLDET_WIDE_WG16": "1"}),
}
for k_, v_ in env_work.items():
    work[k_] = v_
bad_any = 0
for name, item in work.items():
    cfg, C, S, engine = item[:4]
    env = item[4] if len(item) > 4 else {}
    for k_ in ("SYLDET_FUSED_CLASSIC", "SYLDET_FUSED_NOFOLD", "SYLDET_FUSED_NOFOLD2", "SYLDET_WIDE_NOSTAGGER", "SYLDET_WIDE_WG16"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
    with sd.SyllableDetector(cfg, channels=C, engine=engine) as det:
        det.profile(True)
        o0, f0_ = det.run(x)
        torch.cuda.synchronize()
        o0, f0_ = torch.nan_to_num(o0.clone()), f0_.clone()
        names = [n for n, _ in det.lastTimings()]
        bad = 0
        for _ in range(runs - 1):
            o, f = det.run(x)
            torch.cuda.synchronize()
            bad += int(not (torch.equal(torch.nan_to_num(o), o0) and torch.equal(f, f0_)))
        cols_bad = 0
        if engine == _abi.ENGINE_AUTO and not name.startswith("config3"):
            c0 = det.spectrogram(x).clone()
            for _ in range(3):
                cols_bad += int(not torch.equal(det.spectrogram(x), c0))
    print("%-46s %-58s %d of %d runs differ from the first; spectrogram %d of 3" % (name, names, bad, runs - 1, cols_bad), flush=True)
    bad_any += bad + cols_bad
    del x
sys.exit(1 if bad_any else 0)
