#!/bin/bash
# Diagnostic (run on the GPU box via gpurun): utilisation counters of bench.py's kernels, one small group per pass
# (tools/pmc_run.sh: --kernel-trace + --pmc only, guarded by a timeout).  usage: tools/pmc_counters.sh [bench args...]
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
  bash tools/pmc_run.sh $name "$grp" "$@" | grep -A6 "fused_kernel\|wide_gemm" | grep -v "^--"
done
