#!/usr/bin/env python3
"""Diagnostic: GEMM kernel time of the wide-network engine on the benchmark batch for different hidden transfer
functions (PureLin shows the matrix work + pipeline overhead without the transcendental epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import nets, synth, _abi

base = nets.from_npz()
C, S = 64, 1 << 24
x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=base.samplingRate)
for tf in ("TanSig", "LogSig", "SatLin", "PureLin"):
    cfg = nets.wide_mlp(base)
    cfg.net.layers[0].transferFunction = tf
    with sd.SyllableDetector(cfg, channels=C, engine=_abi.ENGINE_WIDE_BF16) as det:
        E = det.countEvaluations(S)
        out = torch.empty((C, E, 1), dtype=torch.float32, device="cuda")
        fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
        det.profile(True)
        ms = []
        for i in range(4):
            det.run(x, out, fl)
            if i >= 1:
                ms.append(next(v for k, v in det.lastTimings() if k.startswith("wide_gemm")))
        t = sum(ms) / len(ms)
        print("%-8s gemm %.2f ms   %.0f TFLOP/s (K = 290)" % (tf, t, C * E * (2 * 290 * 4096 + 2 * 4096) / (t * 1e-3) / 1e12), flush=True)
