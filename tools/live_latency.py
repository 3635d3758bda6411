#!/usr/bin/env python3
"""Live-use round trip: interleaved audio callbacks -> sample rings -> processAll() (one staged copy, one
launch, one copy back for all channels) -> per-channel results.  Prints wall-clock per callback.

    python tools/live_latency.py [channels] [frames_per_callback] [callbacks]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import syllable_detector_swift_amd as sd  # noqa: E402
from syllable_detector_swift_amd import nets, synth  # noqa: E402


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    cfg = nets.from_npz()
    x = np.stack([synth.channel(n * rounds, 1000 + c) for c in range(C)])
    for mode in ("processAll", "per-channel"):
        with sd.SyllableDetector(cfg, channels=C) as det:
            t_app, t_proc, t_drain, evals = [], [], [], 0
            for r in range(rounds):
                blk = np.ascontiguousarray(x[:, r * n:(r + 1) * n].T)
                t0 = time.perf_counter()
                det.appendInterleavedData(blk)
                t1 = time.perf_counter()
                if mode == "processAll":
                    det.processAll()
                t2 = time.perf_counter()
                for c in range(C):
                    while det.processNewValue(c):
                        evals += 1
                t3 = time.perf_counter()
                t_app.append(t1 - t0), t_proc.append(t2 - t1), t_drain.append(t3 - t2)
            w = rounds // 10                                      # skip warm-up callbacks
            tot = np.array(t_proc[w:]) + np.array(t_drain[w:])
            print(json.dumps({"mode": mode, "channels": C, "frames_per_callback": n, "callbacks": rounds,
                              "audio_ms_per_callback": 1e3 * n / cfg.samplingRate, "evaluations": evals,
                              "append_us_median": 1e6 * float(np.median(t_app[w:])),
                              "process_us_median": 1e6 * float(np.median(t_proc[w:])),
                              "drain_us_median": 1e6 * float(np.median(t_drain[w:])),
                              "round_trip_us_median": 1e6 * float(np.median(tot)),
                              "round_trip_us_p99": 1e6 * float(np.percentile(tot, 99))}))


if __name__ == "__main__":
    main()
