#!/bin/bash
# Round profile (run on the GPU box via gpurun, from the repository root): rocprofv3 kernel-trace statistics and PMC passes
# of bench.py for the three workloads, written under gpurun_out/prof_<round>/ for tools/profile_collect.py to condense.
#   usage: tools/profile_round.sh r02
# Counters are collected in passes of their own (--kernel-trace + --pmc only), the program right behind "--".
set -u
R=${1:-r06}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench args
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $ROOT/bench.py --no-cpu-baseline --no-also "$@" > $OUT/stats_$name.log 2>&1
  echo "stats $name rc=$?"
}
run_pmc() {     # name, counters, bench args
  name=$1; ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 3 --warmup 1 --preroll 0 --no-cpu-baseline --no-verify --no-also "$@" > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
# (--stats averages over every dispatch, pre-roll and warmup included: enough timed steps that the ramp's 25 launches weigh
# under 1 %; tools/profile_collect.py also reads the timed region's dispatches out of the trace by themselves)
run_stats sample --steps 500 --warmup 5
run_stats config3 --workload config3 --steps 300 --warmup 5
run_stats config5 --workload config5 --steps 20 --warmup 2
run_pmc sample_fetch FETCH_SIZE
run_pmc sample_write WRITE_SIZE
run_pmc config3_fetch FETCH_SIZE --workload config3
run_pmc config3_write WRITE_SIZE --workload config3
for grp in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  run_pmc sample_SQ_$tag "$grp"
  run_pmc config3_SQ_$tag "$grp" --workload config3
  run_pmc config5_SQ_$tag "$grp" --workload config5
done
cd $ROOT
python3 tools/profile_collect.py $R
