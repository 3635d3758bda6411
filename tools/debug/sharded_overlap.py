#!/usr/bin/env python3
"""Debug: does the exchange of batch i run beside the kernel of batch i + 1 in the one-process sharded bank?  Four batches of
BASELINE configs[3]'s per-GPU shape (512 channels x 2^21 samples) through syldet_sharded_run_device with devices = {0} (one RCCL
rank), queued back to back.  Run it under the profiler and read the trace with --read:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/overlap -- python3 tools/debug/sharded_overlap.py
    python3 tools/debug/sharded_overlap.py --read gpurun_out/overlap"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import csv
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    main = [r for r in rows if "fused_s_kernel" in r[2]]
    unpack = [r for r in rows if "unpack_flags" in r[2]]
    pack = [r for r in rows if "pack_flags" in r[2] and "unpack" not in r[2]]
    t0 = rows[0][0]
    print("%d fused_s_kernel, %d pack, %d unpack dispatches" % (len(main), len(pack), len(unpack)))
    # the timed batches are the last four of each kind
    main, unpack, pack = main[-4:], unpack[-4:], pack[-4:]
    for k in range(len(main)):
        print("batch %d: kernel %9.1f .. %9.1f us   pack %9.1f .. %9.1f   unpack %9.1f .. %9.1f" % (
            k, (main[k][0] - t0) / 1e3, (main[k][1] - t0) / 1e3, (pack[k][0] - t0) / 1e3, (pack[k][1] - t0) / 1e3, (unpack[k][0] - t0) / 1e3, (unpack[k][1] - t0) / 1e3))
    ok = all(main[k + 1][0] < unpack[k][1] for k in range(len(main) - 1))
    print("every batch's kernel starts before the batch before's unpacking ends:", ok)
    sys.exit(0 if ok else 1)

import torch
from syllable_detector_swift_amd import nets, synth
from syllable_detector_swift_amd.bank import ShardedSyllableDetectorBank
cfg = nets.from_npz()
C, S = 512, 1 << 21
x = synth.channels_on_device(C, S, torch.device("cuda", 0))
with ShardedSyllableDetectorBank(cfg, C, [0]) as bank:
    outs, fls, alls = bank.run([x], S)                  # (first call: communicator, buffers)
    bank.synchronize()
    for k in range(4):
        bank.run([x], S, outputs=outs, flags=fls, flags_all=alls)
    bank.synchronize()
print("done")
