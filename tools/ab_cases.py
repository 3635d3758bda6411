#!/usr/bin/env python3
"""Diagnostic: kernel times of the engines BEHIND the benchmark's three -- the 1024-point FFT kernel (a Blackman window keeps it), the
generic engine's two launches (dB columns on the lanes FFT + the matrix-core network; the plain generic pair), the pass-scaled fused
kernels -- under the library SYLDET_LIB names, medians of 20 launches each:   python tools/ab_cases.py
(run it once per library on one box, e.g. through tools/variant_libs.sh builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

base = nets.from_npz()
rng = np.random.default_rng(2)
cases = [
    ("fft1k (configs[2] under a Blackman window)", nets.variant(nets.config3(), window=_abi.WINDOW_BLACKMAN), 256, 1 << 21, _abi.ENGINE_AUTO, {}),
    ("generic engine, example geometry", base, 64, 1 << 22, _abi.ENGINE_GENERIC, {}),
    ("configs[2], generic engine", nets.config3(), 128, 1 << 21, _abi.ENGINE_GENERIC, {}),
    ("8-wave fused kernel", base, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_CLASSIC": "1"}),
    ("register-resident fused kernel", base, 64, 1 << 23, _abi.ENGINE_AUTO, {"SYLDET_FUSED_NOFOLD": "1"}),
    ("H = 16, 4 outputs on the fold kernel", nets.variant(base, net=nets.random_net(rng, 290, (16,), 4), thresholds=[0.5] * 4), 64, 1 << 23, _abi.ENGINE_AUTO, {}),
    ("wide bands (58 bins)", nets.variant(base, freqRange=(1000.0, 11000.0), net=nets.random_net(rng, 58 * 10, (4,), 1)), 64, 1 << 23, _abi.ENGINE_AUTO, {}),
]
for name, cfg, C, S, engine, env in cases:
    for k in ("SYLDET_FUSED_CLASSIC", "SYLDET_FUSED_NOFOLD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
    with sd.SyllableDetector(cfg, channels=C, engine=engine) as det:
        det.profile(True)
        per = {}
        for i in range(26):
            det.run(x)
            if i >= 6:
                for n, ms in det.lastTimings():
                    per.setdefault(n, []).append(ms)
        torch.cuda.synchronize()
    print("%-44s %s" % (name, "  ".join("%s %.4f" % (n, sorted(v)[len(v) // 2]) for n, v in per.items())), flush=True)
    del x
