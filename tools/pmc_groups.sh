#!/bin/bash
# Diagnostic (run on the GPU box via gpurun): utilisation counters of bench.py's dominant kernel, one small group per pass
# (tools/pmc_run.sh: --kernel-trace + --pmc only, guarded by a timeout).  usage: tools/pmc_groups.sh <tag> [bench args...]
tag=$1; shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  name=pmc_${tag}_$(echo $grp | tr ' ' '_' | cut -c1-40)
  bash tools/pmc_run.sh $name "$grp" --no-also --no-verify "$@" 2>&1 | grep -A8 "fused_s_kernel\|bdft_net\|wide_gemm16" | grep -v "^--" | grep -v fixup
done
