#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r4_pmc2
export PMC_SETS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for s in "full up_g4.last" "gru3.h-step B96" "gru1.h-step B96" "pw gru_1x1_4 full" "full g2.first" "tempD.d1.last 3d"; do
  tag=$(echo "$s" | tr ' .' '__')
  EXTRA="" bash tools/pmc_conv.sh "r4_pmc2/$tag" bf16x3 "$s" > "gpurun_out/r4_pmc2/$tag.log" 2>&1
  echo "== $s"; cat "gpurun_out/r4_pmc2/$tag"/pmc_pass*.csv | grep -v "^kernel" | awk -F, '{printf "%s %s %s %s\n", substr($1,30,60), $2, $3, $5}'; grep "fwd" "gpurun_out/r4_pmc2/$tag/p1.log" | tail -1 | cut -c1-110
done
