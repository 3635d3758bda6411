#!/usr/bin/env python3
"""Diagnostic: wall time of the knock-out instantiation of the fused kernel on the benchmark workload with parts knocked out
(SYLDET_FUSED_KO bit mask; results are wrong by construction).  One process per mask."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import torch
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd import nets, synth
    cfg = nets.from_npz()
    C, S = 64, 1 << 24
    det = sd.SyllableDetector(cfg, channels=C, device=0, engine=2)
    x = synth.channels_on_device(C, S, torch.device("cuda", 0), fs=cfg.samplingRate)
    E = det.countEvaluations(S)
    out = torch.empty((C, E, 1), dtype=torch.float32, device="cuda")
    fl = torch.empty((C, E), dtype=torch.uint8, device="cuda")
    sys.stderr = open(os.devnull, "w")
    os.dup2(sys.stderr.fileno(), 2)
    for _ in range(2):
        det.run(x, out, fl)
    torch.cuda.synchronize()
    det.profile(True)
    ms = []
    for _ in range(5):
        det.run(x, out, fl)
        ms.append(det.lastTimings()[0][1])
    print("%.3f ms" % (sum(ms) / len(ms)))
else:
    for ko in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 32, 64, 128, 144, 255]:
        env = dict(os.environ, SYLDET_FUSED_KO=str(ko))
        r = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
        print("KO=%3d  %s" % (ko, r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
