#!/bin/bash
# PMC passes on the dominant conv kernel (run through gpurun): MFMA busy, HBM read, HBM write in SEPARATE rocprofv3 runs
# (TCC slots: FETCH_SIZE costs 3, WRITE_SIZE 2 - MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage: bash tools/pmc_conv.sh <tag> <precision> <shape substring>
#   PMC_TOOL=tools/wgrad_check.py EXTRA="--modes=1,2" profiles the weight-gradient kernels instead; PMC_SETS="a b|c d" replaces the counter sets
set -u
TAG=${1:-pmc}; PREC=${2:-bf16x3}; SHAPE=${3:-up_g4.first T18}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
i=0
SETS=${PMC_SETS:-"SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE|FETCH_SIZE|WRITE_SIZE|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"}
IFS='|' read -r -a SETARR <<< "$SETS"
for CTRS in "${SETARR[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -f csv -d "$OUT/p$i" -o pmc -- \
      python "$ROOT/${PMC_TOOL:-tools/conv_bench.py}" --prec=$PREC ${EXTRA:-} "$SHAPE" > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
  F=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python "$ROOT/tools/pmc_summarize.py" "$F" "$OUT/pmc_pass$i.csv"
  rm -rf "$OUT/p$i"
done
cat "$OUT"/pmc_pass*.csv
