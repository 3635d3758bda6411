#!/usr/bin/env python3
"""Condenses what tools/profile_round.sh wrote under gpurun_out/prof_<round>/ into the small files that are committed under
profiles/: per-kernel statistics (rocprofv3 --stats), HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE: the gfx950
correction of MI355X_MICROARCH.md, HBM), utilisation counters per dispatch, and the bench lines of the profiled runs."""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + R)
DST = os.path.join(ROOT, "gpurun_out", "profiles_" + R)
os.makedirs(DST, exist_ok=True)


def kname(full):
    """'void sd::(anonymous namespace)::fused_r_kernel<8, 12, 9, false, false>(sd::FusedDesc, ...)' -> 'fused_r_kernel'"""
    m = re.search(r"(\w+_kernel)\b", full)
    return m.group(1) if m else full[:40]


def bench_line(log):
    try:
        for l in reversed(open(log).read().splitlines()):
            if l.startswith("{"):
                return json.loads(l)
    except Exception:
        pass
    return None


def kernel_stats(name):
    f = glob.glob(os.path.join(SRC, "stats_" + name, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        return None
    rows = list(csv.DictReader(open(f[0])))
    keep = [r for r in rows if "sd::" in r["Name"]]
    out = os.path.join(DST, "%s_%s_kernel_stats.csv" % (R, name))
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow(r)
    return {kname(r["Name"]): {"calls": int(r["Calls"]), "average_ms": float(r["AverageNs"]) / 1e6} for r in keep}


def timed_region_ms(name, kernel, steps):
    """rocprofv3's own timestamps of the LAST `steps` full-size dispatches of `kernel` -- the bench's timed region (the
    dispatches before them are its pre-roll and warmup steps; `--stats` averages over all of them, the clock ramp included)."""
    f = glob.glob(os.path.join(SRC, "stats_" + name, "**", "*kernel_trace.csv"), recursive=True)
    if not f or not steps:
        return None
    rows = [r for r in csv.DictReader(open(f[0])) if kernel in r["Kernel_Name"]]
    if not rows:
        return None
    grid = collections.Counter((r["Grid_Size_X"], r["Grid_Size_Y"]) for r in rows).most_common(1)[0][0]
    rows = [r for r in rows if (r["Grid_Size_X"], r["Grid_Size_Y"]) == grid]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-steps:]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
    return {"dispatches": len(d), "average_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d), "of_full_size_dispatches": len(rows)}


def counters(name):
    f = glob.glob(os.path.join(SRC, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}
    agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for row in csv.DictReader(open(f[0])):
        if "sd::" not in row["Kernel_Name"]:
            continue
        k = kname(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    return {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in agg.items()}


summary = {"source": "tools/profile_round.sh %s (rocprofv3 --kernel-trace --stats; separate --kernel-trace --pmc passes), MI355X" % R}
for wl, dom in (("sample", "fused_s_kernel"), ("config3", "fft1k_net_kernel"), ("config5", "wide_gemm_kernel")):
    line = bench_line(os.path.join(SRC, "stats_%s.log" % wl))
    if line and line.get("roofline", {}).get("kernel"):
        dom = line["roofline"]["kernel"]                 # what the run says its dominant kernel was
    ks = kernel_stats(wl)
    if line:
        json.dump(line, open(os.path.join(DST, "%s_%s_bench_under_rocprof.json" % (R, wl)), "w"), indent=1)
    summary[wl] = {"kernel_stats": ks, "bench_kernel_ms": line["roofline"]["kernel_ms"] if line else None,
                   "bench_steps": line.get("steps") if line else None, "bench_preroll_steps": line.get("preroll_steps") if line else None,
                   "timed_region_from_the_trace": timed_region_ms(wl, dom, line.get("steps") if line else 0)}
    fetch, write = (counters(wl + "_fetch"), counters(wl + "_write")) if wl != "config5" else ({}, {})    # (config5's roof is the matrix pipe)
    if dom in fetch and dom in write and line:
        fk, wk = fetch[dom]["FETCH_SIZE"], write[dom]["WRITE_SIZE"]
        hbm = 2.0 * fk * 1024.0 + wk * 1024.0
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        t = {"workload": {"channels_per_gpu": line["config"]["channels_per_gpu"], "samples_per_channel": line["config"]["samples_per_channel"],
                          "hop": line["config"]["hop"], "engine": line["config"]["engine"]},
             "kernel": dom,
             "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/profile_round.sh), python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify%s, per dispatch of %s" % ("" if wl == "sample" else " --workload " + wl, dom),
             "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
             "correction": "gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads: x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
             "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg}
        json.dump(t, open(os.path.join(DST, "%s_%straffic.json" % (R, "" if wl == "sample" else wl + "_")), "w"), indent=1)
        summary[wl]["traffic_ratio"] = hbm / alg
    util = {}
    for f in glob.glob(os.path.join(SRC, "pmc_%s_SQ*" % wl)):          # (every utilisation group's directory is pmc_<workload>_SQ_<counters>)
        if os.path.isdir(f):
            for k, d in counters(os.path.basename(f)[4:]).items():
                util.setdefault(k, {}).update(d)
    if dom in util:
        c = util[dom]
        der = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            der["mfma_pipe_busy_fraction (MFMA_BUSY / (4 SIMDs x BUSY_CU))"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
        if "SQ_LDS_IDX_ACTIVE" in c and "SQ_BUSY_CU_CYCLES" in c:
            der["lds_active_fraction (LDS_IDX_ACTIVE / BUSY_CU)"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
            der["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            der["wait_any_fraction_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
            der["wait_inst_any_fraction_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            der["valu_to_mfma_instruction_ratio"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
        # The decomposition that closes: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so /8 is the launch in
        # shader clocks; the matrix pipe's busy fraction BY TIME is instructions x 16 clocks (a 16x16x32 MFMA is four passes of
        # four) over 1024 SIMDs x those clocks -- SQ_BUSY_CU_CYCLES, the old denominator, shrinks when CUs idle at the launch's
        # tail and flatters the pipe.  roofline = busy x useful K / padded K x sclk / 2.4 GHz for the wide GEMM.
        tr = summary[wl].get("timed_region_from_the_trace") or {}
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
            clocks = c["GRBM_GUI_ACTIVE"] / 8.0
            der["launch_shader_clocks (GRBM_GUI_ACTIVE / 8 XCDs)"] = clocks
            ms = (ks or {}).get(dom, {}).get("average_ms") or tr.get("average_ms")
            if ms:
                der["sclk_GHz (clocks / rocprofv3 average duration)"] = clocks / (ms * 1e6)
            if "SQ_INSTS_MFMA" in c:
                busy = c["SQ_INSTS_MFMA"] * 16.0 / (1024.0 * clocks)
                der["mfma_pipe_busy_fraction_by_time (INSTS_MFMA x 16 / (1024 SIMDs x clocks))"] = busy
                if wl == "config5" and ms:
                    der["roofline_from_counters (busy x 290/320 x sclk / 2.4 GHz)"] = busy * 290.0 / 320.0 * (clocks / (ms * 1e6)) / 2.4
            if "SQ_INSTS_VALU" in c:
                der["valu_issue_slots_used_fraction (INSTS_VALU x 4 / (1024 SIMDs x clocks); packed and transcendental instructions take 8)"] = c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * clocks)
        summary[wl]["utilisation"] = {"kernel": dom, "counters": c, "derived": der}
json.dump(summary, open(os.path.join(DST, "%s_profile_summary.json" % R), "w"), indent=1)
print(json.dumps({k: (v if k == "source" else {kk: vv for kk, vv in v.items() if kk != "utilisation"}) for k, v in summary.items()}, indent=1)[:3000])
