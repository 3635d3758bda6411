#!/bin/bash
# Diagnostic (run on the GPU box from the repository root): where the block-transform kernel's LDS bank conflicts come from -- the
# LDS counters of BASELINE configs[2] under knock-out builds of kernels_bdft.hip (one stage removed each; results wrong by
# construction), one rocprofv3 counter pass a build.  Builds: tools/variant_libs.sh kernels_bdft.hip ko_<name> "-fno-slp-vectorize -DSYLDET_B_NO<...>"
#   usage: tools/bdft_conflicts.sh name[:libdir] ...      (name "shipped" = the shipped library)
ROOT=$PWD
OUT=$ROOT/gpurun_out/bdft_conflicts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}
  if [ "$name" = "shipped" ]; then unset SYLDET_LIB; else export SYLDET_LIB=$ROOT/syllable_detector_swift_amd/lib_$name/libsyldet.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/$name -- \
    python3 $ROOT/bench.py --workload config3 --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline --no-verify --no-also > $OUT/$name.log 2>&1
  echo "$name rc=$?"
done
cd $ROOT
python3 - "$@" <<'PY'
import csv, glob, os, sys
out = os.path.join(os.getcwd(), "gpurun_out", "bdft_conflicts")
print("%-14s %14s %14s %10s %12s %10s" % ("build", "conflict cyc", "LDS active", "conflict %", "LDS insts", "busy CU"))
for spec in sys.argv[1:]:
    name = spec.split(":")[0]
    tot = {}
    n = 0
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bdft_net_kernel" not in r["Kernel_Name"]:
                continue
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n += 1
    launches = max(1, n // max(1, len(tot)))
    g = lambda k: tot.get(k, 0.0) / launches
    print("%-14s %14.3e %14.3e %10.1f %12.3e %10.3e" % (name, g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE"), 100.0 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0), g("SQ_INSTS_LDS"), g("SQ_BUSY_CU_CYCLES")))
PY
