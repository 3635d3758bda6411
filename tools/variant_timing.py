#!/usr/bin/env python3
"""Diagnostic: kernel time of the fused engine on the benchmark batch for network shapes outside the compile-time-exact
instantiation (wider hidden layer, several outputs, other input chains)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

base = nets.from_npz()
C, S = 64, 1 << 24
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=base.samplingRate)
rng = np.random.default_rng(1)
cases = {
    "sample.txt (LEAN)": base,
    "H=8, 1 output": nets.variant(base, net=nets.random_net(rng, 290, (8,), 1)),
    "H=16, 4 outputs": nets.variant(base, net=nets.random_net(rng, 290, (16,), 4), thresholds=[0.5] * 4),
    "H=4, normalize chain": nets.variant(base, net=nets.random_net(rng, 290, (4,), 1, in_fns=("normalize",))),
    "H=4, no chain, LogSig": nets.variant(base, net=nets.random_net(rng, 290, (4,), 1, transfer=("LogSig", "PureLin"), in_fns=())),
    "T=12": nets.variant(base, timeRange=12, net=nets.random_net(rng, 29 * 12, (4,), 1)),
    "T=4": nets.variant(base, timeRange=4, net=nets.random_net(rng, 29 * 4, (4,), 1)),
    "T=8, hop 100": nets.variant(base, timeRange=8, windowOverlap=156, net=nets.random_net(rng, 29 * 8, (4,), 1)),
    "hop 128": nets.variant(base, windowOverlap=128),
    "H=4, 2 outputs": nets.variant(base, net=nets.random_net(rng, 290, (4,), 2), thresholds=[0.5] * 2),
    "H=4, 4 outputs": nets.variant(base, net=nets.random_net(rng, 290, (4,), 4), thresholds=[0.5] * 4),
    "H=4, normalizestd chain": nets.variant(base, net=nets.random_net(rng, 290, (4,), 1, in_fns=("normalizestd", "mapstd"))),
    "H=6, 1 output": nets.variant(base, net=nets.random_net(rng, 290, (6,), 1)),
    "H=8, T=8": nets.variant(base, timeRange=8, net=nets.random_net(rng, 29 * 8, (8,), 1)),
    "H=8, T=12": nets.variant(base, timeRange=12, net=nets.random_net(rng, 29 * 12, (8,), 1)),
    "H=8, T=10, hop 120": nets.variant(base, windowOverlap=136, net=nets.random_net(rng, 290, (8,), 1)),
    "H=8, T=5": nets.variant(base, timeRange=5, net=nets.random_net(rng, 29 * 5, (8,), 1)),
    "H=8, T=11": nets.variant(base, timeRange=11, net=nets.random_net(rng, 29 * 11, (8,), 1)),
    "H=8, N=128 hop 64": nets.variant(base, fourierLength=128, windowLength=128, windowOverlap=64, freqRange=(2000.0, 6900.0), net=nets.random_net(rng, 15 * 10, (8,), 1)),
    "H=8, hop 64": nets.variant(base, windowOverlap=192, net=nets.random_net(rng, 290, (8,), 1)),
    "H=8, hop 68": nets.variant(base, windowOverlap=188, net=nets.random_net(rng, 290, (8,), 1)),
    "H=8, hop 128": nets.variant(base, windowOverlap=128, net=nets.random_net(rng, 290, (8,), 1)),
    "H=8, hop 96": nets.variant(base, windowOverlap=160, net=nets.random_net(rng, 290, (8,), 1)),
}
# shapes outside the fused engine (AUTO: generic FFT + whichever network stage applies)
auto_cases = {
    "N=512 hop 128, 58 bins": nets.variant(base, fourierLength=512, windowLength=512, windowOverlap=384, net=nets.random_net(rng, 58 * 10, (4,), 1, in_fns=("l2normalize", "mapminmax"))),
    "N=512 hop 256, 58 bins, dB": nets.variant(base, fourierLength=512, windowLength=512, windowOverlap=256, spectrogramScaling="db", net=nets.random_net(rng, 58 * 10, (4,), 1, in_fns=("l2normalize", "mapminmax"))),
    "N=128 hop 64, 15 bins": nets.variant(base, fourierLength=128, windowLength=128, windowOverlap=64, net=nets.random_net(rng, 15 * 10, (4,), 1, in_fns=("l2normalize", "mapminmax"))),
    "sample.txt, dB": nets.variant(base, spectrogramScaling="db"),
    "sample.txt, H=12": nets.variant(base, net=nets.random_net(rng, 290, (12,), 1, in_fns=("l2normalize", "mapminmax"))),
}
if "--short" in sys.argv:            # short frames and hops (AUTO)
    def short(N, ov, bins=15, T=10):
        return nets.variant(base, fourierLength=N, windowLength=N, windowOverlap=ov, freqRange=(2000.0, 2000.0 + (bins - 0.6) * base.samplingRate / N),
                            net=nets.random_net(rng, bins * T, (4,), 1, in_fns=("l2normalize", "mapminmax")))
    auto_cases = {"N=128 hop 64": short(128, 64), "N=128 hop 68": short(128, 60), "N=128 hop 72": short(128, 56), "N=128 hop 96": short(128, 32),
                  "N=256 hop 64": short(256, 192, 29), "N=256 hop 68": short(256, 188, 29), "N=256 hop 96": short(256, 160, 29), "N=64 hop 32": short(64, 32, 8)}
    if "--long" in sys.argv:         # long hops: little or no overlap, gaps between frames
        auto_cases = {"N=256 hop 144": short(256, 112, 29), "N=256 hop 160": short(256, 96, 29), "N=256 hop 192": short(256, 64, 29), "N=256 hop 256": short(256, 0, 29),
                      "N=256 hop 320 (gap 64)": short(256, -64, 29), "N=128 hop 128": short(128, 0), "N=128 hop 192 (gap 64)": short(128, -64)}
    sys.argv.append("--auto")
if "--auto" in sys.argv:
    cases = auto_cases
if "--only" in sys.argv:
    key = sys.argv[sys.argv.index("--only") + 1]
    cases = {k: v for k, v in cases.items() if key in k}
for name, cfg in cases.items():
    with sd.SyllableDetector(cfg, channels=C, engine=_abi.ENGINE_AUTO if "--auto" in sys.argv else _abi.ENGINE_FUSED) as det:
        E = det.countEvaluations(S)
        out = torch.empty((C, E, det.geometry.outputs), dtype=torch.float32, device="cuda")
        fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
        det.profile(True)
        ms = []
        for i in range(14):
            det.run(x, out, fl)
            if i >= 4:
                ms.append(det.lastTimings()[0][1])
        J = det.countFrames(S)
        torch.cuda.synchronize()
        names = "+".join(n for n, _ in det.lastTimings())
        if "--auto" in sys.argv:        # several kernels a step: the sum of the last step's
            tot = sum(t for _, t in det.lastTimings())
            print("%-28s %-60s %.3f ms   %.3g frames/s" % (name, " + ".join("%s %.2f" % (n, t) for n, t in det.lastTimings()), tot, C * J / (tot * 1e-3)), flush=True)
            continue
        print("%-26s %-16s %.3f ms   %.3g frames/s   guard work items %d" % (name, det.lastTimings()[0][0], sum(ms) / len(ms), C * J / (sum(ms) / len(ms) * 1e-3), det.fixupStats()[0]), flush=True)
