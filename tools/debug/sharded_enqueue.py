#!/usr/bin/env python3
"""Debug: what ONE host thread spends queueing a batch of the one-process sharded bank (per shard: the kernel, the packing, the
waits and records of the exchange, its copies or collective, the unpacking -- under hipSetDevice switching), against the 0.87 ms
the benchmark's kernel takes.  Eight shards on the one GPU of a test box (devices = {0} x 8, copy exchange), a tiny batch so that
the device is never the limit; also one shard (RCCL, one rank).

    python tools/debug/sharded_enqueue.py [--shards 8] [--batches 200]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from syllable_detector_swift_amd import nets, synth, _abi
from syllable_detector_swift_amd.bank import ShardedSyllableDetectorBank

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--batches", type=int, default=200)
a = ap.parse_args()
cfg = nets.from_npz()
for shards, exchange, label in ((a.shards, _abi.EXCHANGE_PEER_COPY, "copy exchange"), (1, _abi.EXCHANGE_RCCL, "RCCL, one rank")):
    C, S = 8 * shards, 256 + 132 * 40
    x = synth.channels(C, S, first=5)
    with ShardedSyllableDetectorBank(cfg, C, [0] * shards, exchange=exchange) as bank:
        blocks = bank.scatter(x)
        outs, fls, alls = bank.run(blocks, S)
        bank.synchronize()
        print("launcher threads: %d%s" % (bank.launcherThreads, " (SYLDET_SHARDED_INLINE)" if os.environ.get("SYLDET_SHARDED_INLINE") else ""), flush=True)
        fls_b = [torch.empty_like(f) for f in fls]
        prepared = (bank.prepare(blocks, S, outs, fls, alls), bank.prepare(blocks, S, outs, fls_b, alls))
        for form in ("prepared call (the ABI call alone)", "bank.run (argument checks in Python every call)"):
            for gather in (True, False):
                if form.startswith("prepared") and not gather:
                    continue
                t = []
                for k in range(a.batches):
                    t0 = time.perf_counter()
                    if form.startswith("prepared"):
                        prepared[k & 1]()
                    else:
                        bank.run(blocks, S, gather=gather, outputs=outs, flags=fls, flags_all=alls if gather else None)
                    t.append(time.perf_counter() - t0)
                    if k % 16 == 15:
                        bank.synchronize()
                bank.synchronize()
                t = np.sort(np.array(t)) * 1e3
                print("%d shards on one device, %s, %s, %s: host time per batch call median %.3f ms, p90 %.3f ms" % (
                    shards, label, "with the exchange" if gather else "kernels only", form, t[len(t) // 2], t[int(0.9 * len(t))]), flush=True)
