#!/usr/bin/env python3
"""Headline benchmark: audio frames/sec of the STFT + network hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (samples in HBM -> network outputs + detection flags in HBM) over one batch of
synthetic audio.

N = 1: BASELINE.json configs[1] -- the sample.txt network (N=W=256, overlap 124 => hop 132, Hamming, bins [12,41), T=10,
290 -> 4 TanSig -> 1 PureLin), 64 synthetic channels x 2^24 samples (4 GiB, far past the 256 MiB Infinity Cache), fp32.

N > 1 (launched by torch.distributed.run, one rank per GPU): BASELINE.json configs[3] -- the same network, 512 channels x
2^21 samples per GPU (4096 channels over 8 GPUs; the same 4 GiB and the same 8.13 M frames per GPU as the N = 1 batch), weak
scaling.  Channels are independent detectors (Processor.swift:57-59, main.swift:86-89), so the data path has no collective;
the timed step ends with ONE RCCL all-gather of the per-channel detection flags, as bits.  `--total-channels 4096` fixes
the total instead and shards it over the ranks (strong scaling, dist.shard_channels).

Before the W warmup steps the same step runs `--preroll` more times untimed (default 150, about 0.15 s; `preroll_steps` in the
line): from an idle device the clock governor needs 30-50 ms of launches to reach the state it then holds for seconds
(profiles/r03_clock_ramp.txt), and W + K = 25 launches would otherwise be timed inside that ramp.  The timed region is K steps.

`from_idle` in the line is the same K-step measurement taken FIRST, before any pre-roll (W warmup launches from an idle device,
then K timed steps): both regimes are in the record.

`python bench.py --gpus N` WITHOUT a torch.distributed launcher (no WORLD_SIZE in the environment), or with
`--single-process`, drives all N GPUs from ONE process through the library's sharded bank (syldet_create_sharded: a sub-bank and a
stream per device, every shard launched before any is waited for, one ncclAllGather of the bit-packed flags inside the
library's own RCCL group) -- the shape of the reference, which is one process that owns every channel
(Processor.swift:57-59,128-141) -- and prints the same line with `rccl_ranks` = N and `launcher` = "single-process".

Rank 0 prints one JSON line.  `roofline` is computed from HIP-event timings of the dominant kernel taken inside the timed
region on the launch stream.  After the timed region the last step's results are spot-checked against the CPU oracle
(tests/spotcheck.py: head, tail and segment seams of the first, a middle and the last channel) -> `verified`.
`cpu_baseline` is the oracle's fp32 port of the reference's call sequence in its streaming, frame-at-a-time form, built
-O3 -march=native on this host, single-threaded like the reference (plus an all-core figure beside it), on a bounded sample
of the same workload (rank 0, N = 1 only).
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The host driver of these boxes supports dmabuf IPC only: without this RCCL (and any sharing of device memory between processes)
# fails with `hipIpcGetMemHandle: invalid argument`.  It is exported on the pool's boxes already; set here, before anything loads the
# HIP runtime, for a launcher that starts ranks from a cleaner environment.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)


def cpu_baseline(cfg, samples_host, budget_s=10.0, threads=1):
    """The oracle's fp32 port (kind "port") in the reference's own form: one detector per thread, samples appended in
    8192-sample buffers, one frame and one evaluation at a time through the two rings (orc_stream_run; TrackDetector.swift:
    45-105), built -O3 -march=native on this host.  Every thread works through its channel over and over for ~budget_s
    (ctypes releases the GIL inside the C call); the figure is the median over passes of frames / pass time, times threads."""
    import concurrent.futures as cf
    import pyoracle as po
    po.lib(native=True)                                  # builds here, outside the timed loops
    S = samples_host.shape[1]
    t0 = time.perf_counter()

    def worker(k):
        o = po.Oracle(po.from_config(cfg))
        J = o.count_frames(S)
        rates = []
        while True:
            t1 = time.perf_counter()
            o.stream_run(samples_host[k % samples_host.shape[0]], po.F32, chunk=8192, native=True, keep=False)
            rates.append(J / (time.perf_counter() - t1))
            if time.perf_counter() - t0 >= budget_s and len(rates) >= 5:
                return rates

    with cf.ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(worker, range(threads)))
    dt = time.perf_counter() - t0
    passes = sum(len(r) for r in res)
    J = po.Oracle(po.from_config(cfg)).count_frames(S)
    # one thread: the median pass rate (the reference is single-threaded per detector bank); several threads: all frames of all
    # passes over the wall time of the whole leg, stragglers and idle tails included -- with the sum of per-thread medians beside it
    value = statistics.median(res[0]) if threads == 1 else passes * J / dt
    return {"value": value, "unit": "frames/s", "cores": threads, "kind": "port", "form": "streaming, frame at a time (since round 2; round 1 timed the batch form)",
            "sum_of_per_thread_median_rates": sum(statistics.median(r) for r in res),
            "sample": "%d thread(s) x 1 channel x %d samples of the benchmark input, streaming frame-at-a-time form "
                      "(8192-sample buffers, one processNewValue per evaluation), median of %d passes (%.1f s), oracle fp32 "
                      "port at -O3 -march=native (radix-2 packed real FFT, unfolded network)" % (threads, S, passes, dt)}


def measured_traffic(C, S, hop, engine, kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r*_traffic*.json: FETCH_SIZE
    and WRITE_SIZE collected separately, gfx950 FETCH_SIZE x2 correction), newest round first, if they were taken on
    exactly this workload and kernel; else (None, None).  The run itself does not collect counters."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            t = json.load(open(path))
            w = t["workload"]
            if (w["channels_per_gpu"], w["samples_per_channel"], w["hop"], w["engine"]) == (C, S, hop, engine) and \
                    (kernel is None or t.get("kernel", kernel) == kernel):
                return t["hbm_bytes_per_launch"], os.path.relpath(path, ROOT) + " (rocprofv3 --pmc passes of this command, committed; not collected by this run)"
        except Exception:
            pass
    return None, None


# What the builder's evidence says binds each dominant kernel (the HBM / MFMA fraction stays the contract's yardstick): the
# resource, where the evidence is, and -- measured by this run -- the socket power and shader clock while the kernel loops.
BINDING = {
    "fused_s_kernel": ("power+latency", "looped for seconds the launch sits on the 1400 W limit with the shader clock at 1.9-2.1 GHz "
                       "(tools/power_probe.py; round 4's 1175 W was two samples of a 0.2 s run: the package figure lags); in a run as short "
                       "as the timed region the clock is still high and what is left is two waves a SIMD both waiting inside one tile's "
                       "dependent chain, ~68 % of the vector issue slots in use; HBM traffic 1.01x algorithmic (MEASUREMENTS R4.1, R5.3)"),
    "bdft_net_kernel": ("valu_issue+power", "~240 vector and 24 matrix instructions a wave and 16-frame iteration (a tile's end runs inside the next tile since "
                        "round 5): two waves a SIMD fill most vector issue slots, near the part's power limit; HBM traffic 1.00x "
                        "algorithmic (MEASUREMENTS R3.5, R4.5, R5.4)"),
    "fft1k_net_kernel": ("valu_issue+lds", "radix-8 register FFT with two LDS transposes a frame (MEASUREMENTS, old 4.2c)"),
    "wide_gemm16_kernel": ("mfma+valu_issue", "matrix pipe busy ~60 % of the launch's clocks: the epilogue's two transcendentals a hidden value "
                           "share the issue port with the matrix instructions' issue, at 1290-1350 W with the clock below 2.4 GHz; "
                           "K = 290 padded to 320 costs 9.4 % of every MFMA (MEASUREMENTS R4.6, R5.1, R5.5)"),
}


def binding_of(kernel, launch, seconds=1.0, device_index=0):
    """`launch()` queues one step.  Loops it for `seconds` while the package's power and shader-clock sensors are read (hwmon, every
    10 ms): socket power and clock under THIS kernel (the timed region is tens of milliseconds: the package figure lags behind it).  Issue-slot use comes from the committed PMC
    passes when they cover the kernel (profiles/r*_profile_summary.json), like `traffic`."""
    import threading
    import torch
    resource, evidence = BINDING.get(kernel, ("unknown", "no evidence recorded for this kernel"))
    stop = [False]

    # the package's own sensors through sysfs (what rocm-smi prints): no child process, nothing executed beside a process that holds the GPU
    # WHICH package: the card whose PCI address is this HIP device's (a box shows one GPU of a host of eight: card0 is not it in
    # general); without a match every package is read and the one that draws most during the loop is taken, and the record says so
    hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
    hwdirs = [os.path.dirname(h) for h in hw]
    sensor = None
    try:
        import ctypes
        buf = ctypes.create_string_buffer(64)
        if ctypes.CDLL("libamdhip64.so").hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
            bdf = buf.value.decode().lower()                       # "0000:c5:00.0"
            for hd in hwdirs:
                if bdf and bdf in os.path.realpath(os.path.join(hd, "..", "..")).lower():
                    hwdirs, sensor = [hd], "pci " + bdf
                    break
    except Exception:
        pass

    def read_num(hd, name):
        try:
            return float(open(os.path.join(hd, name)).read().strip())
        except Exception:
            return None
    per = {hd: ([], []) for hd in hwdirs}

    def poll():
        while not stop[0] and hwdirs:
            for hd in hwdirs:
                w, f = read_num(hd, "power1_input"), read_num(hd, "freq1_input")
                if w is not None:
                    per[hd][0].append(w / 1e6)
                if f is not None:
                    per[hd][1].append(f / 1e6)
            time.sleep(0.01)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            launch()
        torch.cuda.synchronize()
        n += 4
    looped_ms = 1e3 * (time.perf_counter() - t0) / max(n, 1)
    stop[0] = True
    th.join(timeout=6)
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    hwdir = max(hwdirs, key=lambda hd: med(per[hd][0]) or 0.0) if hwdirs else None
    if hwdir and sensor is None:
        sensor = "busiest of %d packages" % len(hwdirs)
    watts, mhz = per[hwdir] if hwdir else ([], [])
    cap = read_num(hwdir, "power1_cap") if hwdir else None
    rec = {"resource": resource, "evidence": evidence, "socket_w": med(watts[len(watts) // 2:]), "sclk_mhz": med(mhz[len(mhz) // 2:]),   # (the loop's second half: the ramp is over)
           "power_limit_w": (cap / 1e6) if cap else 1400,
           "samples": len(watts), "sensor": sensor, "launches_while_sampled": n,
           "ms_per_step_looped": looped_ms,        # (wall clock over the loop, a synchronisation every four steps: the regime of a job that runs for seconds)
           "issue_slot_utilisation": None, "issue_slot_source": None}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_profile_summary.json")), reverse=True):
        try:
            for wl, v in json.load(open(path)).items():
                u = v.get("utilisation") if isinstance(v, dict) else None
                if u and u.get("kernel") == kernel:
                    for k, val in u.get("derived", {}).items():
                        if k.startswith("valu_issue_slots_used_fraction"):
                            rec["issue_slot_utilisation"] = val
                            rec["issue_slot_source"] = os.path.relpath(path, ROOT) + " (SQ_INSTS_VALU x 4 clocks / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): a floor, packed and transcendental instructions hold the port for 8)"
                        if k.startswith("mfma_pipe_busy_fraction_by_time"):
                            rec["mfma_pipe_busy_by_time"] = val
            if rec["issue_slot_utilisation"] is not None:
                break
        except Exception:
            pass
    return rec


def side_record(workload, local_rank, steps=20, warmup=3, verify=True, preroll_s=0.15):
    """One sub-record of the `also` object: another single-GPU BASELINE workload (configs[2] "config3": 1024-point frames;
    configs[4] "config5": the 4096-hidden network as a bf16 MFMA GEMM) or the headline workload on adversarial audio
    ("clicks": a full-scale click every 64 frames over a cage at -80 dBFS, which the precision guard legitimately sends to
    the exact fp64 recomputation; "hop128": the same network on 256-point frames at hop 128, BASELINE configs[0]'s framing --
    SURVEY 8(d) names the variant -- where every frame starts on the same LDS banks), measured in this process after the headline's timed region: the same launch loop, the
    kernels' own HIP events, the oracle spot-check of the last step."""
    import torch
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd import nets, synth
    dev = torch.device("cuda", local_rank)
    engine = 0
    if workload == "config3":
        cfg, C, S = nets.config3(), 512, 1 << 21
    elif workload == "config5":
        cfg, C, S, engine = nets.wide_mlp(nets.from_npz()), 64, 1 << 24, 3
    else:
        cfg, C, S = nets.from_npz(), 64, 1 << 24
        if workload == "hop128":                                     # BASELINE configs[0]'s framing (256-point frames, hop 128) at configs[1]'s size
            cfg = nets.variant(cfg, windowOverlap=128)
    with sd.SyllableDetector(cfg, channels=C, device=local_rank, engine=engine) as det:
        g = det.geometry
        J, E = det.countFrames(S), det.countEvaluations(S)
        x = synth.channels_on_device(C, S, dev, fs=cfg.samplingRate)
        if workload == "clicks":
            x.mul_(1e-4 / synth.NOISE_RMS)                       # the cage: noise at -80 dBFS (bursts at -70)
            x[:, 1000::64 * g.hop] = 1.0                         # a full-scale click every 64 frames
        outputs = torch.empty((C, E, g.outputs), dtype=torch.float32, device=dev)
        flags = torch.empty((C, E), dtype=torch.uint8, device=dev)
        det.profile(True, history=steps)
        tp = time.perf_counter()                                     # (the clock ramp: see the headline's pre-roll)
        while time.perf_counter() - tp < preroll_s:
            for _ in range(5):
                det.run(x, outputs, flags)
            torch.cuda.synchronize()
        for _ in range(warmup):
            det.run(x, outputs, flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            det.run(x, outputs, flags)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = {}
        for back in range(steps):
            for nm, ms in det.timingsOf(back):
                kernel_ms.setdefault(nm, []).append(ms)
        means = {k: sum(v) / len(v) for k, v in kernel_ms.items()}
        fixups, overflow = det.fixupStats()
        rec = {"workload": workload, "value": C * J * steps / elapsed, "unit": "frames/s", "steps": steps, "warmup": warmup, "preroll_s": preroll_s,
               "ms_per_step": 1e3 * elapsed / steps, "channels": C, "samples_per_channel": S, "frames_per_channel": J,
               "fixups": {"work_items_last_step": fixups, "overflow": overflow}}
        if engine == 3:
            L = cfg.net.layers
            f_frame = 2 * L[0].inputs * L[0].outputs + 2 * L[0].outputs * L[1].outputs
            wk = next(k for k in means if k.startswith("wide_gemm"))          # wide_gemm16_kernel, or wide_gemm_kernel under SYLDET_WIDE_SHAPE32
            tf = C * E * f_frame / (means[wk] * 1e-3) / 1e12
            tf_step = C * E * f_frame / (elapsed / steps) / 1e12        # end to end: the spectrogram front's launch is inside the step
            rec["roofline"] = {"bound": "mfma", "kernel": wk, "kernel_ms": means[wk], "achieved": tf,
                               "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TFLOPS,
                               "frac_end_to_end": tf_step / MFMA_BF16_PEAK_TFLOPS, "achieved_end_to_end": tf_step,
                               "note": "frac: the GEMM kernel's launch from its HIP events; frac_end_to_end: the same flops over the whole step (spectrogram front + GEMM)",
                               "all_kernels_ms": means}
        else:
            dom = max((k for k in means if k != "fixup_kernel"), key=means.get)
            b_frame = 4 * g.hop + 4 * g.outputs + 1
            gbs = C * J * b_frame / (means[dom] * 1e-3) / 1e9
            gbs_step = C * J * b_frame / (elapsed / steps) / 1e9
            traffic, traffic_source = (None, None) if workload == "clicks" else \
                measured_traffic(C, S, g.hop, {1: "generic", 2: "fused"}.get(g.engine, ""), None)
            rec["roofline"] = {"bound": "hbm", "kernel": dom, "kernel_ms": means[dom], "achieved": gbs, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frac_by_step": gbs_step / HBM_PEAK_GBS,
                               "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": C * J * b_frame, "all_kernels_ms": means}
        wk_or_dom = rec["roofline"]["kernel"]
        rec["roofline"]["kernel_ms_per_step"] = [round(v, 4) for v in reversed(kernel_ms[wk_or_dom])]   # (in launch order)
        try:
            rec["roofline"]["binding"] = binding_of(wk_or_dom, lambda: det.run(x, outputs, flags), seconds=0.8)
        except Exception as e:
            rec["roofline"]["binding"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if verify:
            import spotcheck
            try:
                v = spotcheck.check(det, cfg, x, outputs, flags, sorted({0, max(C // 2 - 1, 0), C - 1}),
                                    width=160 if workload != "config3" else 64, tol=1e-2 if engine == 3 else 1e-5)
                rec["verified"] = True
                rec["verify"] = {k: v[k] for k in ("evaluations_checked", "detections", "max_error", "tolerance")}
                # once more on planted syllables (an extra launch after the timed region): flags that fire, compared exactly
                chs = sorted({0, max(C // 2 - 1, 0), C - 1})
                if workload == "hop128" and spotcheck.plant(det, cfg, x, chs):
                    det.run(x, outputs, flags)
                    torch.cuda.synchronize()
                    v = spotcheck.check(det, cfg, x, outputs, flags, chs, tol=1e-5)
                    rec["verify_planted"] = {k: v[k] for k in ("evaluations_checked", "detections", "max_error", "tolerance")}
                    rec["verified"] = bool(v["detections"] > 0)
            except AssertionError as e:
                rec["verified"] = False
                rec["verify"] = {"error": str(e)[:300]}
        del x, outputs, flags
    torch.cuda.empty_cache()
    return rec


def live_record(local_rank, channels=64, verify=True):
    """The reference's real use (N4): an audio callback appends a few frames per channel (32-frame buffers:
    AudioInterface.swift:342,474; 512 for comparison), the consumer drains every detector (Processor.swift:102-149).  Here:
    syldet_append_interleaved + syldet_process_all (one staged copy, one launch, one copy back for all channels) + the
    per-channel hand-out; wall time per callback, and the outputs of three channels against the oracle's streaming form."""
    import numpy as np
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd import nets, synth
    import pyoracle as po
    import util
    cfg = nets.from_npz()
    rec = {"channels": channels, "path": "syldet_append_interleaved -> syldet_process_all -> syldet_process_new_value / syldet_last_outputs per channel"}
    for n, rounds in ((32, 3000), (512, 600)):
        S = n * rounds
        x = np.stack([synth.channel(S, 1000 + c) for c in range(channels)])
        got = {c: [] for c in (0, channels // 2, channels - 1)}
        with sd.SyllableDetector(cfg, channels=channels, device=local_rank) as det:
            t_cb, evals = [], 0
            t_start = time.perf_counter()
            for r in range(rounds):
                blk = np.ascontiguousarray(x[:, r * n:(r + 1) * n].T)
                t0 = time.perf_counter()
                det.appendInterleavedData(blk)
                det.processAll()
                for c in range(channels):
                    while det.processNewValue(c):
                        evals += 1
                        if c in got:
                            got[c].append(det.lastOutputsFor(c))
                t_cb.append(time.perf_counter() - t0)
            wall = time.perf_counter() - t_start
        w = rounds // 10
        t = np.array(t_cb[w:])
        hop = cfg.windowLength - cfg.windowOverlap
        r_ = {"frames_per_callback": n, "callbacks": rounds, "audio_us_per_callback": 1e6 * n / cfg.samplingRate,
              "callback_us_median": 1e6 * float(np.median(t)), "callback_us_p99": 1e6 * float(np.percentile(t, 99)),
              "evaluations": evals, "sustained_frames_per_s": channels * (S // hop) / wall,
              "real_time_factor": (S / cfg.samplingRate) / wall}
        if verify:
            o = util.oracle_for(cfg)
            worst = 0.0
            for c, rows in got.items():
                want, _ = o.stream_run(x[c], po.F64, chunk=n)
                a = np.array(rows, np.float32).reshape(-1, want.shape[1])
                assert a.shape == want.shape, (a.shape, want.shape)
                worst = max(worst, float(np.abs(a.astype(np.float64) - want).max()))
            r_["verified"] = bool(worst <= 1e-5)
            r_["max_error_vs_oracle_streaming_form"] = worst
        rec["callbacks_of_%d" % n] = r_
    return rec


def host_record(cfg, det_engine, local_rank, x):
    """The boundary on host buffers (syldet_run: H2D, kernel, D2H pipelined along time inside the library) on a bounded sample,
    ordinary and page-locked memory, beside the box's own page-locked H2D rate: PCIe, not the kernel, bounds this form."""
    import numpy as np
    import torch
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd.bank import PinnedArray
    hc, hs = 16, 1 << 23                                           # 512 MiB of input: two stages of the pipeline
    hx = x[:hc, :hs].cpu().numpy()
    rec = {"unit": "frames/s", "sample": "%d channels x %d samples through syldet_run (PCIe both ways; staged along time in 256 MiB stages)" % (hc, hs)}
    with sd.SyllableDetector(cfg, channels=hc, device=local_rank, engine=det_engine) as hdet:
        J, E = hdet.countFrames(hs), hdet.countEvaluations(hs)
        b_frame = 4 * hdet.geometry.hop + 4 * hdet.geometry.outputs + 1
        out = np.zeros((hc, E, hdet.geometry.outputs), np.float32)
        fl = np.zeros((hc, E), np.uint8)
        hdet.runHost(hx, out, fl)
        t1 = time.perf_counter()
        for _ in range(3):
            hdet.runHost(hx, out, fl)
        rec["value"] = 3 * hc * J / (time.perf_counter() - t1)
        px, po_, pf = PinnedArray(hx.shape, np.float32), PinnedArray(out.shape, np.float32), PinnedArray(fl.shape, np.uint8)
        px.array[:] = hx
        hdet.runHost(px.array, po_.array, pf.array)
        t1 = time.perf_counter()
        for _ in range(3):
            hdet.runHost(px.array, po_.array, pf.array)
        rec["page_locked_buffers"] = 3 * hc * J / (time.perf_counter() - t1)
        rec["results_equal_pageable"] = bool(np.array_equal(po_.array, out) and np.array_equal(pf.array, fl))
        # the box's page-locked H2D rate on the same bytes (one plain copy), and what it allows at B_frame bytes a frame
        d = torch.empty(hx.shape, dtype=torch.float32, device=torch.device("cuda", local_rank))
        src = torch.from_numpy(px.array)
        d.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            d.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        h2d = 3 * hx.nbytes / (time.perf_counter() - t1)
        rec["h2d_page_locked_GBps"] = h2d / 1e9
        rec["pcie_bound_frames_per_s"] = h2d / (4 * hdet.geometry.hop)
        rec["fraction_of_pcie_bound"] = {"pageable": rec["value"] / rec["pcie_bound_frames_per_s"],
                                         "page_locked": rec["page_locked_buffers"] / rec["pcie_bound_frames_per_s"]}
        rec["algorithmic_bytes_per_frame"] = b_frame
        for p_ in (px, po_, pf):
            p_.free()
    return rec


def open_sharded_bank(cfg, total, devices, engine):
    """The one-process bank under the RCCL exchange, its communicators brought up NOW (syldet_sharded_connect) so that a host whose
    RCCL does not come up is known before anything is timed; on an error the bank is made again IN THIS PROCESS with the copy
    exchange (hipMemcpyPeerAsync, no RCCL) and the line says so.  A process that has touched the GPU is never re-executed."""
    from syllable_detector_swift_amd import _abi
    from syllable_detector_swift_amd.bank import ShardedSyllableDetectorBank
    info = {"exchange": None, "rccl_error": None}
    bank = ShardedSyllableDetectorBank(cfg, total, devices, engine=engine)
    try:
        bank.connect()
    except Exception as e:
        info["rccl_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        bank.close()
        bank = ShardedSyllableDetectorBank(cfg, total, devices, engine=engine, exchange=_abi.EXCHANGE_PEER_COPY)
        bank.connect()
    info["exchange"] = "rccl" if bank.rcclRanks > 0 else "peer_copy"
    return bank, info


def single_process(args, devices=None, emit=True, tag="BASELINE configs[3]"):
    """All N GPUs from ONE process through the sharded bank (BASELINE configs[3]'s shape per GPU; weak scaling).  `devices`: the
    HIP device of every shard (default 0 .. N-1; a device listed several times rehearses the whole branch on one GPU, with the copy
    exchange -- RCCL refuses duplicates)."""
    import numpy as np
    import torch
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd import _abi, nets, synth
    from syllable_detector_swift_amd.bank import ShardedSyllableDetectorBank
    devices = list(devices) if devices is not None else list(range(args.gpus))
    N = len(devices)
    json_fd = None
    if emit:
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    cfg = nets.from_npz()
    if args.overlap is not None:
        cfg = nets.variant(cfg, windowOverlap=args.overlap)
    Cg = args.channels or 512
    S = 1 << (args.log2_samples or 21)
    total = args.total_channels if args.total_channels is not None else N * Cg
    scaling = "strong" if args.total_channels is not None else "weak"
    preroll = args.preroll if args.preroll is not None else 150

    def build(bank):
        g = bank.geometry
        E = bank.countEvaluations(S)
        blocks, outs, fls, fls_b, alls = [], [], [], [], []
        for i, sh in enumerate(bank.shards):
            dev = torch.device("cuda", sh.device)
            blocks.append(synth.channels_on_device(sh.channels, S, dev, first=sh.first_channel, fs=cfg.samplingRate))
            outs.append(torch.empty((sh.channels, E, g.outputs), dtype=torch.float32, device=dev))
            fls.append(torch.empty((sh.channels, E), dtype=torch.uint8, device=dev))
            fls_b.append(torch.empty((sh.channels, E), dtype=torch.uint8, device=dev))
            alls.append(torch.empty((total, E), dtype=torch.uint8, device=dev))
        # The bank's streams are its own (non-blocking): they do not wait for torch's stream, on which the audio above is still
        # being generated -- and the result tensors just handed out may be the generator's freed temporaries (torch's allocator
        # reuses them in ITS stream's order).  Everything torch queued finishes before the bank's first kernel is queued.
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        # every shard's own flags land in one of two tensors in turn: a batch's flags are packed on the exchange stream while the
        # next batch's kernel runs, and a kernel that writes the tensor still being packed from would have to wait for that (the
        # library checks).  Two prepared calls (arguments checked once; a call is then the ABI call alone).
        calls = (bank.prepare(blocks, S, outs, fls, alls), bank.prepare(blocks, S, outs, fls_b, alls))
        calls[1]()
        calls[0]()                                                   # the first gathering batches: buffers, and under RCCL the first collective
        bank.synchronize()
        return blocks, outs, fls, fls_b, alls, calls

    bank, xinfo = open_sharded_bank(cfg, total, devices, args.engine)
    try:
        blocks, outs, fls, fls_b, alls, calls = build(bank)
    except Exception as e:
        if bank.rcclRanks == 0:
            raise
        # the communicators came up but the first collective did not go through: the copy exchange, in this same process
        xinfo["rccl_error"] = "first batch: %s: %s" % (type(e).__name__, str(e)[:300])
        bank.close()
        torch.cuda.empty_cache()
        bank = ShardedSyllableDetectorBank(cfg, total, devices, engine=args.engine, exchange=_abi.EXCHANGE_PEER_COPY)
        bank.connect()
        xinfo["exchange"] = "peer_copy"
        blocks, outs, fls, fls_b, alls, calls = build(bank)
    g = bank.geometry
    J, E = bank.countFrames(S), bank.countEvaluations(S)
    dets = [sd.SyllableDetector.borrowed(bank, i) for i in range(N)]
    for d in dets:
        d.profile(True, history=max(args.steps, 1))

    left = [preroll + args.warmup + args.steps]                   # steps still to come: the last one (0 left after it) writes `fls`
    host_s = []

    def step():
        left[0] -= 1
        t = time.perf_counter()
        calls[0 if left[0] % 2 == 0 else 1]()
        host_s.append(time.perf_counter() - t)

    for _ in range(preroll + args.warmup):
        step()
    bank.synchronize()
    del host_s[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    bank.synchronize()
    elapsed = time.perf_counter() - t0
    host_ms = sorted(1e3 * v for v in host_s)
    b_frame = 4 * g.hop + 4 * g.outputs + 1
    per_dev = []
    for d in dets:
        km = {}
        for back in range(args.steps):
            for nm, ms in d.timingsOf(back):
                km.setdefault(nm, []).append(ms)
        per_dev.append({k: sum(v) / len(v) for k, v in km.items()})
    dom = max((k for k in per_dev[0] if k != "fixup_kernel"), key=per_dev[0].get)
    worst_ms = max(m[dom] for m in per_dev)
    fix = [d.fixupStats() for d in dets]
    C0 = bank.shards[0].channels
    Cmax = max(sh.channels for sh in bank.shards)
    step_ms = 1e3 * elapsed / args.steps
    by_events = C0 * J * b_frame / (per_dev[0][dom] * 1e-3) / 1e9
    by_step = Cmax * J * b_frame / (step_ms * 1e-3) / 1e9          # per device: the longest shard's bytes over the whole job's step
    value = total * J * args.steps / elapsed
    distinct = len(set(devices))
    line = {"metric": "audio frames/sec (256-pt STFT + 2-layer MLP), whole job", "value": value, "unit": "frames/s", "n_gpus": distinct,
            "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32 (I/O and accumulate; products as f16 hi/lo split x3 on MFMA)" if g.engine == 2 else "f32", "data": "synthetic",
            "per_gpu": value / distinct, "launcher": "single-process", "shards": N, "devices": devices,
            "config": {"workload": "%s: sample.txt network, %d channels sharded across %d MI355X (%d x 2^%d samples per shard), %s of detection flags"
                                   % (tag, total, distinct, Cg, S.bit_length() - 1, "RCCL gather" if xinfo["exchange"] == "rccl" else "peer-copy gather"),
                       "channels_per_gpu": [sh.channels for sh in bank.shards], "total_channels": total, "samples_per_channel": S, "frames_per_channel": J,
                       "evaluations_per_channel": E, "fourier_length": cfg.fourierLength, "hop": g.hop, "bins": [g.f0, g.f1], "time_range": cfg.timeRange,
                       "engine": {1: "generic", 2: "fused", 3: "wide_bf16"}.get(g.engine, str(g.engine)),
                       "sharding": "one process, one handle (syldet_create_sharded): a sub-bank, two streams and a launcher thread per device, contiguous channel "
                                   "blocks, no data-path collective; one all-gather of the bit-packed flags per step (ncclAllGather inside the library's own "
                                   "RCCL group, ncclCommInitAll; hipMemcpyPeerAsync under the copy exchange)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": by_step, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by_step / HBM_PEAK_GBS,
                         "achieved_kernel_events": by_events, "frac_kernel_events": by_events / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": C0 * J * b_frame, "algorithmic_bytes_per_frame": b_frame,
                         "kernel_ms": per_dev[0], "kernel_ms_per_device": [m[dom] for m in per_dev], "slowest_device_kernel_ms": worst_ms,
                         "note": "per device; frac: the longest shard's algorithmic bytes over the job's step time; frac_kernel_events: device 0's launch from its HIP events"},
            "exchange": xinfo["exchange"], "rccl_error": xinfo["rccl_error"], "rccl_ranks": bank.rcclRanks, "launcher_threads": bank.launcherThreads,
            "host_enqueue_ms": {"median": host_ms[len(host_ms) // 2], "p90": host_ms[int(0.9 * (len(host_ms) - 1))], "max": host_ms[-1],
                                "note": "wall time of the one call that queues a batch on every shard (the kernel takes ~0.9 ms)"},
            "gathered_flags_shape": [total, E],
            "gathered_bytes_per_rank_per_step": Cmax * ((E + 7) // 8),
            "fixups": {"work_items_last_step_per_shard": [int(f[0]) for f in fix], "overflow": int(max(f[1] for f in fix)),
                       "note": "16-evaluation items the precision guard sent to the exact fp64 recomputation (0 for ordinary audio)"}}
    if not args.no_verify:
        # every device holds every channel's flags: all copies equal device 0's, whose own rows equal its local flags; and the
        # last step's outputs of the first and last shard against the oracle -- then once more on planted syllables (an extra,
        # untimed step), so that the flags compared are not all zero
        import spotcheck

        def gathered_ok():
            same = all(bool(torch.equal(alls[i].cpu(), alls[0].cpu())) for i in range(1, N))
            own = all(bool(torch.equal(alls[i][sh.first_channel: sh.first_channel + sh.channels], fls[i])) for i, sh in enumerate(bank.shards))
            return bool(same and own)
        ok = gathered_ok()
        line["gathered_flags_identical_on_every_device"] = ok
        try:
            checks = {}
            for i in sorted({0, N - 1}):
                sh = bank.shards[i]
                torch.cuda.set_device(sh.device)
                checks["shard%d" % i] = spotcheck.check(dets[i], cfg, blocks[i], outs[i], fls[i], sorted({0, sh.channels - 1}), tol=1e-5)
            line["verify"] = checks
            planted = 0
            for i in sorted({0, N - 1}):
                planted += spotcheck.plant(dets[i], cfg, blocks[i], sorted({0, bank.shards[i].channels - 1}))
            if planted:
                calls[0]()
                bank.synchronize()
                ok2 = gathered_ok()
                pchecks = {"gathered_flags_identical_on_every_device": ok2}
                for i in sorted({0, N - 1}):
                    sh = bank.shards[i]
                    torch.cuda.set_device(sh.device)
                    pchecks["shard%d" % i] = spotcheck.check(dets[i], cfg, blocks[i], outs[i], fls[i], sorted({0, sh.channels - 1}), tol=1e-5)
                pchecks["detections"] = sum(v["detections"] for k, v in pchecks.items() if k.startswith("shard"))
                line["verify_planted"] = pchecks
                ok = ok and ok2 and pchecks["detections"] > 0
            line["verified"] = bool(ok)
        except AssertionError as e:
            line["verified"] = False
            line["verify"] = {"error": str(e)[:400]}
    line["summary"] = {"frac_by_step": line["roofline"]["frac"], "frac_by_kernel_events": line["roofline"]["frac_kernel_events"],
                       "n_gpus": distinct, "shards": N, "exchange": xinfo["exchange"], "rccl_ranks": bank.rcclRanks,
                       "host_enqueue_ms_median": line["host_enqueue_ms"]["median"], "verified": line.get("verified"),
                       "detections_checked": line.get("verify_planted", {}).get("detections")}
    bank.close()
    del blocks, outs, fls, fls_b, alls, calls
    torch.cuda.empty_cache()
    if emit:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preroll", type=int, default=None,
                    help="untimed launches of the step before the warmup steps, to bring an idle device's clocks to their sustained state (default 150; 0: from cold)")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: 64 at one GPU, 512 at several)")
    ap.add_argument("--log2-samples", type=int, default=None, help="samples per channel = 2^k (default: 24 at one GPU, 21 at several)")
    ap.add_argument("--total-channels", type=int, default=None, help="strong scaling: this many channels sharded over the ranks")
    ap.add_argument("--overlap", type=int, default=None, help="override windowOverlap (128 => the hop-128 variant)")
    ap.add_argument("--workload", default="sample", choices=["sample", "config3", "config5"])
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 generic, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` sub-records (configs[2], configs[4], clicks over quiet audio)")
    ap.add_argument("--force-gather", action="store_true", help="run the flag exchange even with one rank (rehearsal of the multi-GPU step)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from this one process through the library's sharded bank (the default when --gpus > 1 and no torch.distributed launcher set WORLD_SIZE)")
    ap.add_argument("--devices", default=None,
                    help="single-process form: the HIP device of every shard, e.g. 0,1,2,3; a device listed several times (0,0,0,0,0,0,0,0) rehearses the "
                         "whole N > 1 branch on one GPU with the copy exchange")
    args = ap.parse_args()
    if args.devices is not None:
        devs = [int(v) for v in args.devices.split(",") if v.strip() != ""]
        args.gpus = len(devs)
        single_process(args, devices=devs)
        return
    if args.single_process or (args.gpus > 1 and "WORLD_SIZE" not in os.environ):
        single_process(args)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import syllable_detector_swift_amd as sd
    from syllable_detector_swift_amd import nets, synth
    from syllable_detector_swift_amd.dist import HostFlagGather, PipelinedFlagGather, gather_flags, shard_channels

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    exchange = world > 1 or args.force_gather
    # stdout carries ONE line, the JSON: libraries that write there from native code (RCCL prints a version banner when its
    # first communicator comes up) are sent to stderr for the length of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # one rank per GPU: LOCAL_RANK's device (modulo the devices this process sees: two ranks on a one-GPU box rehearse the whole
    # branch -- RCCL refuses two ranks on one device, which is exactly the failure the fallback below is for)
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    nccl_group, xinfo = None, {"exchange": None, "rccl_error": None}
    if exchange:
        torch.cuda.set_device(local_dev)
        # The job's control plane is a gloo group (rendezvous, barriers, the reductions of the timing): it comes up wherever TCP to
        # 127.0.0.1 does.  RCCL carries the one data exchange, on a group of its own -- brought up and PROBED here, before anything is
        # timed, and every rank learns over gloo whether every rank's probe went through: either all ranks exchange over RCCL or all
        # fall back to the host-staged gloo gather (dist.HostFlagGather), in this same process, and the line says which.
        if world == 1:                                   # rehearsal: one rank
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29593")
            dist.init_process_group(backend="gloo", rank=0, world_size=1)
        else:
            dist.init_process_group(backend="gloo")
        ok = 1
        try:
            if os.environ.get("SYLDET_BENCH_NO_RCCL"):                 # (test hook: the fallback on a box where RCCL works)
                raise RuntimeError("SYLDET_BENCH_NO_RCCL is set")
            nccl_group = dist.new_group(backend="nccl")
            probe = torch.zeros((world, 8), dtype=torch.uint8, device=torch.device("cuda", local_dev))
            mine = torch.full((8,), rank + 1, dtype=torch.uint8, device=torch.device("cuda", local_dev))
            dist.all_gather_into_tensor(probe, mine, group=nccl_group)
            torch.cuda.synchronize()
            if probe.cpu().tolist() != [[r + 1] * 8 for r in range(world)]:
                raise RuntimeError("the probe all-gather returned wrong rows")
        except Exception as e:
            ok = 0
            xinfo["rccl_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        agreed = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            nccl_group = None
            if xinfo["rccl_error"] is None:
                xinfo["rccl_error"] = "another rank's RCCL probe failed"
        xinfo["exchange"] = "rccl" if nccl_group is not None else "gloo_host"
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)
    local_rank = local_dev                               # (every later use is "this rank's device")

    multi = exchange                                     # the multi-GPU step (or its one-rank rehearsal): configs[3]'s shape
    if args.workload == "config3":
        cfg = nets.config3()
        C = args.channels or 512
        S = 1 << (args.log2_samples or 21)
        name = "BASELINE configs[2]: 1024-pt FFT hop 256, 512 channels, synthetic 1160-4-1 network"
    elif args.workload == "config5":
        cfg = nets.wide_mlp(nets.from_npz())
        C = args.channels or 64
        S = 1 << (args.log2_samples or 24)
        name = "BASELINE configs[4]: sample.txt front end, 290 -> 4096 TanSig -> 1 network, bf16 MFMA GEMM"
        if args.engine == 0:
            args.engine = 3                      # the wide engine is opt-in (bf16 numerics)
    else:
        cfg = nets.from_npz()
        C = args.channels or (512 if multi else 64)
        S = 1 << (args.log2_samples or (21 if multi else 24))
        name = ("BASELINE configs[3]: sample.txt network, 4096 channels sharded across 8 MI355X (512 x 2^21 samples per GPU), "
                "RCCL gather of detection flags" if multi else "BASELINE configs[1]: sample.txt network, 64 synthetic channels on 1 MI355X, 256-pt FFT, Hamming")
    if args.overlap is not None:
        cfg = nets.variant(cfg, windowOverlap=args.overlap)
    scaling, first = "weak", rank * C
    total = world * C
    if args.total_channels is not None:                  # strong scaling: a fixed bank sharded over the ranks
        scaling, total = "strong", args.total_channels
        first, C = shard_channels(total, world, rank)
        assert C > 0, "fewer channels than ranks: dist.ShardedSyllableDetector shards the time axis then; this benchmark does not"

    det = sd.SyllableDetector(cfg, channels=C, device=local_rank, engine=args.engine)
    g = det.geometry
    J, E = det.countFrames(S), det.countEvaluations(S)
    x = synth.channels_on_device(C, S, dev, first=first, fs=cfg.samplingRate)
    outputs = torch.empty((C, E, g.outputs), dtype=torch.float32, device=dev)
    flags = torch.empty((C, E), dtype=torch.uint8, device=dev)
    # every kernel of every timed step bracketed by HIP events on the library's stream, all of them kept and read after the
    # timed region: the loop below never waits for a step before launching the next
    det.profile(True, history=max(args.steps, 1))
    # ONE collective per batch: [total, E] u8 flags on every rank, as bits; with equal shards the exchange of batch i runs on
    # a side stream under the kernel of batch i+1 (the timed region ends with every exchange finished: synchronize below)
    equal = total == world * C
    gather = None
    if exchange and equal:
        gather = PipelinedFlagGather(C, E, total, dev, group=nccl_group) if nccl_group is not None else HostFlagGather(C, E, total, dev)
    gathered = None

    # (with the pipelined exchange the flags land in two tensors in turn: a batch's flags are packed on the side stream while the
    # next batch's kernel runs; `flags` holds the last step's -- the one verified below -- and those of every second one before it)
    flags_b = torch.empty_like(flags) if gather is not None else None
    left = [0]                                           # steps still to come in the current stretch (set by the loops below)

    def step():
        nonlocal gathered
        if gather is not None:
            left[0] -= 1
            fl = flags if left[0] % 2 == 0 else flags_b
            gather.before_run(fl)
            det.run(x, outputs, fl)
            gather.submit(fl)
            return
        det.run(x, outputs, flags)
        if exchange:
            # ragged shards: the padded form, on the compute stream (through the host when RCCL did not come up)
            gathered = gather_flags(flags, total, group=nccl_group) if nccl_group is not None else gather_flags(flags.cpu(), total, packed=False).to(dev)

    # From an idle device the clock governor takes 30-50 ms of back-to-back launches to reach the state it then holds for
    # seconds (tools/ramp_probe.py, profiles/r03_clock_ramp.txt: 1.42, 1.17, 1.04, 0.977 ms a launch over the first 50 launches of
    # this batch, 0.973-0.975 from there to 6 s) -- W + K = 25 launches would be timed inside that ramp.  A pre-roll of the same
    # step (the same count on every rank: the step may hold a collective) brings the device there first; the W warmup steps
    # and the K timed steps follow unchanged.  --preroll 0 times from cold.
    preroll = args.preroll if args.preroll is not None else (150 if args.engine != 3 and args.workload != "config5" else 10)
    # ... and BEFORE it, the from-idle regime for the record: W warmup launches on the idle device, then K timed steps
    # (what `--preroll 0` measures), with the same barriers; the pre-rolled measurement below is the headline.
    from_idle = None
    left[0] = ((args.warmup + args.steps) if preroll > 0 else 0) + preroll + args.warmup + args.steps    # every step() to come
    if preroll > 0:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ti = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        idle_elapsed = time.perf_counter() - ti
        idle_ms = {}
        for back in range(args.steps):
            for nm, ms in det.timingsOf(back):
                idle_ms.setdefault(nm, []).append(ms)
        from_idle = (idle_elapsed, {k: sum(v) / len(v) for k, v in idle_ms.items()})
    for _ in range(preroll):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for back in range(args.steps):                       # the timed steps' kernels, from their events
        for nm, ms in det.timingsOf(back):
            kernel_ms.setdefault(nm, []).append(ms)
    frames_local = torch.tensor([float(C * J)], dtype=torch.float64)       # (host tensors: the control plane is the gloo group)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(frames_local, op=dist.ReduceOp.SUM)
    frames_per_step = int(frames_local.item())
    value = frames_per_step * args.steps / elapsed
    fixups, fix_overflow = det.fixupStats()

    if rank == 0:
        # algorithmic bytes per frame: unique input bytes + fp32 outputs + 1 flag byte
        b_frame = 4 * g.hop + 4 * g.outputs + 1
        means = {k: sum(v) / len(v) for k, v in kernel_ms.items()}
        dom = max(means, key=means.get)
        # the dominant kernel's launch covers C*J frames of this rank; when the path is split over
        # several kernels each is charged the whole frame's algorithmic bytes (none moves fewer)
        achieved_events = C * J * b_frame / (means[dom] * 1e-3) / 1e9
        # (the line's `frac` is the conservative one: the same bytes over the STEP time the line reports -- launch gaps and the small
        # kernels behind the dominant one included; the dominant kernel's own launch, from its HIP events, beside it)
        achieved = C * J * b_frame / (elapsed / args.steps) / 1e9
        traffic, traffic_source = measured_traffic(C, S, g.hop, {1: "generic", 2: "fused"}.get(g.engine, ""), dom)
        engine_name = {1: "generic", 2: "fused", 3: "wide_bf16"}.get(g.engine, str(g.engine))
        line = {
            "metric": {"sample": "audio frames/sec (256-pt STFT + 2-layer MLP), whole job",
                       "config3": "audio frames/sec (1024-pt STFT + 2-layer MLP), whole job",
                       "config5": "audio frames/sec (256-pt STFT + 4096-hidden MLP as bf16 MFMA GEMM), whole job"}[args.workload],
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None,
            # what the path computes in: fp32 in and out, fp32 accumulation; the DFT and first-layer products run on the matrix
            # cores as three f16 x f16 products of hi/lo-split operands (2^-22 relative) under power-of-two block scales
            "dtype": "f32 (I/O and accumulate; products as f16 hi/lo split x3 on MFMA)" if g.engine == 2 else "f32",
            "data": "synthetic",
            "per_gpu": value / world,
            "config": {"workload": name, "channels_per_gpu": C, "total_channels": total, "samples_per_channel": S, "frames_per_channel": J,
                       "evaluations_per_channel": E, "fourier_length": cfg.fourierLength, "hop": g.hop,
                       "bins": [g.f0, g.f1], "time_range": cfg.timeRange, "engine": engine_name,
                       "sharding": ("contiguous channel blocks, %d on this rank of %d; no data-path collective; one all-gather of flags (as bits) per step%s"
                                    % (C, total, ", on a side stream under the next step's kernel" if gather is not None else "")) if exchange else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "achieved_kernel_events": achieved_events, "frac_kernel_events": achieved_events / HBM_PEAK_GBS,
                         "frac_is": "algorithmic bytes per launch / ms_per_step (the step the driver times); frac_kernel_events: / the dominant kernel's mean HIP-event duration over the same K launches",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": C * J * b_frame,
                         "algorithmic_bytes_per_frame": b_frame, "kernel_ms": means},
            "fixups": {"work_items_last_step": fixups, "overflow": fix_overflow,
                       "note": "16-evaluation items the precision guard sent to the exact fp64 recomputation (0 for ordinary audio)"},
        }
        if from_idle is not None:
            idle_elapsed, idle_means = from_idle
            idom = max(idle_means, key=idle_means.get)
            line["from_idle"] = {"ms_per_step": 1e3 * idle_elapsed / args.steps, "kernel": idom, "kernel_ms": idle_means[idom],
                                 "frac": C * J * b_frame / (idle_elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                                 "frac_kernel_events": C * J * b_frame / (idle_means[idom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "value_local": C * J * args.steps / idle_elapsed,
                                 "note": "the same K steps timed first, after W warmup launches on an idle device and before the pre-roll (rank 0's clock): what a job of a few dozen launches sees"}
            if g.engine != 3:
                line["roofline"]["from_idle_frac"] = line["from_idle"]["frac"]            # (inside the object the driver keeps)
                line["roofline"]["from_idle_frac_kernel_events"] = line["from_idle"]["frac_kernel_events"]
        if exchange:
            line["launcher"] = "torch.distributed.run, one process per GPU"
            line["rccl_ranks"] = dist.get_world_size() if nccl_group is not None else 0
            line["exchange"], line["rccl_error"] = xinfo["exchange"], xinfo["rccl_error"]
            line["control_plane"] = "gloo (rendezvous, barriers, timing reductions); the flags' all-gather alone is on RCCL"
            line["gathered_flags_shape"] = [total, E]
            line["gathered_bytes_per_rank_per_step"] = C * ((E + 7) // 8) if gather is not None else C * E
        if g.engine == 3:
            # the wide engine's roof is the bf16 matrix pipe: flops of the two layers per evaluation
            L = cfg.net.layers
            f_frame = 2 * L[0].inputs * L[0].outputs + 2 * L[0].outputs * L[1].outputs
            wk = next(k for k in means if k.startswith("wide_gemm"))
            tf = C * E * f_frame / (means[wk] * 1e-3) / 1e12
            line["dtype"] = "bf16"
            tf_step = C * E * f_frame / (elapsed / args.steps) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": wk, "achieved": tf_step, "peak": MFMA_BF16_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": tf_step / MFMA_BF16_PEAK_TFLOPS, "achieved_kernel_events": tf, "frac_kernel_events": tf / MFMA_BF16_PEAK_TFLOPS,
                                "frac_is": "flops per step / ms_per_step (spectrogram front + GEMM: end to end); frac_kernel_events: / the GEMM kernel's own HIP-event duration",
                                "traffic": None, "traffic_source": None,
                                "algorithmic_flops_per_launch": C * E * f_frame, "algorithmic_flops_per_frame": f_frame, "kernel_ms": means}
        if world == 1:
            # what binds the dominant kernel, beside the contract's yardstick (after the timed region: the probe loops the step for a second)
            try:
                line["roofline"]["binding"] = binding_of(line["roofline"]["kernel"], lambda: det.run(x, outputs, flags))
            except Exception as e:
                line["roofline"]["binding"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if not args.no_verify:
            # the last timed step's results against the oracle's fp64 anchor where a tiling bug would show (tests/spotcheck.py);
            # the bf16 engine at its own, separately stated bar
            import spotcheck
            try:
                chs = sorted({0, max(C // 2 - 1, 0), C - 1})
                line["verify"] = spotcheck.check(det, cfg, x, outputs, flags, chs, tol=1e-2 if g.engine == 3 else 1e-5)
                line["verified"] = True
            except AssertionError as e:
                line["verified"] = False
                line["verify"] = {"error": str(e)[:400]}
        if world == 1 and not args.no_cpu_baseline:
            # the boundary also takes host buffers (syldet_run): H2D + kernel + D2H pipelined along time, bounded sample.
            # Reported beside the headline, never as `value`.
            try:
                line["host_buffers"] = host_record(cfg, args.engine, local_rank, x)
            except Exception as e:
                line["host_buffers"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            # (the 4096-unit network costs the CPU 2.4 MFLOP an evaluation: a sixteenth of the sample keeps five passes in the budget)
            n_cpu = min(S, 1 << (18 if g.engine == 3 else 22))
            host = x[:min(C, 8), :n_cpu].cpu().numpy()
            # the reference runs every detector on one serial queue (Processor.swift:82,128; main.swift:126-130): 1 thread
            line["cpu_baseline"] = cpu_baseline(cfg, host, threads=1)
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            ncpu = os.cpu_count() or 1
            if ncpu > 1:                                      # also channels spread over all host cores
                line["cpu_baseline_all_cores"] = cpu_baseline(cfg, host, threads=ncpu)
        if not args.no_verify and line.get("verified") and g.engine != 3:
            # The benchmark's audio never fires the sample network (its flags are all zero): template syllables are written over the
            # checked stretches AFTER the timed region (and after the CPU legs, which read the same tensor), one more (untimed)
            # launch of this rank's kernel alone -- no exchange -- and the same check again: flags that fire, bit for bit
            import spotcheck
            try:
                chs = sorted({0, max(C // 2 - 1, 0), C - 1})
                if spotcheck.plant(det, cfg, x, chs):
                    det.run(x, outputs, flags)
                    torch.cuda.synchronize()
                    line["verify_planted"] = spotcheck.check(det, cfg, x, outputs, flags, chs, tol=1e-5)
                    line["verified"] = bool(line["verify_planted"]["detections"] > 0)
            except AssertionError as e:
                line["verified"] = False
                line["verify_planted"] = {"error": str(e)[:400]}
        if world == 1 and not args.no_also and args.workload == "sample" and args.engine == 0 and args.overlap is None \
                and args.channels is None and args.log2_samples is None and not exchange:
            # the other single-GPU BASELINE workloads and the guard's worst ordinary case, in the same process and on the same
            # box, after the headline's timed region (which they therefore cannot disturb)
            det.close()
            del x, outputs, flags
            torch.cuda.empty_cache()
            line["also"] = {}
            for wl in ("config3", "config5", "clicks", "hop128"):
                try:
                    # (every side record in the headline's own regime -- an idle device, the pre-roll, warmup, K steps -- not in
                    # whatever state the record before it left the package in: configs[4]'s one-second loop at 1350 W does not cool in 0.1 s)
                    time.sleep(1.0)
                    line["also"][wl] = side_record(wl, local_rank, verify=not args.no_verify)
                except Exception as e:                       # a side record must never cost the headline its line
                    line["also"][wl] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            try:
                line["also"]["live"] = live_record(local_rank, verify=not args.no_verify)
            except Exception as e:
                line["also"]["live"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if "roofline" in line["also"].get("clicks", {}):
                line["also"]["clicks"]["slowdown_vs_headline"] = line["also"]["clicks"]["ms_per_step"] / line["ms_per_step"]
            # BASELINE configs[3]'s shape PER GPU (512 channels x 2^21 samples) through the one-process bank on this one device, a
            # one-rank RCCL group carrying the exchange: the like-for-like N = 1 point of the 1 -> 8 curve (`--gpus N` runs this
            # shape on every GPU; the headline above is configs[1]'s 64 x 2^24, whose launch differs by a few per cent)
            try:
                a4 = argparse.Namespace(gpus=1, steps=args.steps, warmup=args.warmup, preroll=100, channels=512, log2_samples=21,
                                        total_channels=None, overlap=None, engine=0, no_verify=args.no_verify)
                r4 = single_process(a4, devices=[local_rank], emit=False, tag="BASELINE configs[3]'s shard shape on one GPU")
                line["also"]["config4_shard"] = {k: r4[k] for k in ("value", "unit", "ms_per_step", "steps", "preroll_steps", "config", "roofline", "exchange", "rccl_error",
                                                                   "rccl_ranks", "launcher_threads", "host_enqueue_ms", "verified", "verify_planted") if k in r4}
            except Exception as e:
                line["also"]["config4_shard"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # LAST key of the line (the driver's record keeps the line's tail): every fraction of the run in one small object
        also = line.get("also", {})
        rf = lambda wl, key="frac": also.get(wl, {}).get("roofline", {}).get(key)
        bind = line["roofline"].get("binding", {}) if isinstance(line["roofline"].get("binding"), dict) else {}
        looped = None
        if bind.get("ms_per_step_looped") and g.engine != 3:
            looped = C * J * b_frame / (bind["ms_per_step_looped"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        line["summary"] = {"workload": args.workload, "n_gpus": world, "value": value,
                           "frac_by_step": line["roofline"]["frac"], "frac_by_kernel_events": line["roofline"]["frac_kernel_events"],
                           "from_idle_frac": line.get("from_idle", {}).get("frac"), "looped_frac": looped,
                           "config3_frac": rf("config3"), "config3_frac_by_step": rf("config3", "frac_by_step"),
                           "config5_frac": rf("config5"), "config5_frac_end_to_end": rf("config5", "frac_end_to_end"),
                           "hop128_frac": rf("hop128"), "hop128_frac_by_step": rf("hop128", "frac_by_step"), "clicks_frac": rf("clicks"),
                           "config4_shard_frac_by_step": rf("config4_shard"), "config4_shard_frac_by_kernel_events": rf("config4_shard", "frac_kernel_events"),
                           "config4_shard_value": also.get("config4_shard", {}).get("value"),
                           "verified": {k: v for k, v in [("headline", line.get("verified"))] + [(wl, r.get("verified")) for wl, r in also.items() if isinstance(r, dict) and "verified" in r]},
                           "detections_checked": line.get("verify_planted", {}).get("detections"),
                           "traffic_over_algorithmic": (line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"])
                           if line["roofline"].get("traffic") and line["roofline"].get("algorithmic_bytes_per_launch") else None,
                           "cpu_baseline_frames_per_s": line.get("cpu_baseline", {}).get("value")}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    det.close()
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
