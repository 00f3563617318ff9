#!/bin/bash
# Round 5: counters for EVERY conv class.
#   A. one rocprofv3 --pmc pass over a WHOLE training step (bench.py, paper config, B = 16): matrix-pipe busy cycles, instruction
#      mix and LDS conflicts of every dispatch -> the time-weighted MFMA-busy share over all 3x3 launches of the real step
#      (tools/pmc_classes.py), per (kernel instantiation, grid).
#   B. the same counters + FETCH_SIZE + WRITE_SIZE (separate passes - TCC slots) on one launch per MODE through tools/conv_bench.py:
#      plain bn96 / bn128 / bn48, phase bn96 / bn128, pooled (data gradient of the upsampling convs), ConvGRU step rows, the
#      wave-specialised weight gradient (plain and by phases).
# --pmc is never combined with a sys/hip/hsa trace (gpurun refuses that); --kernel-trace only supplies the durations.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r5_pmc}
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
SETA="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ "${SKIP_STEP:-0}" != "1" ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc $SETA -f csv -d "$OUT/step" -o pmc -- \
      python "$ROOT/bench.py" --steps 1 --warmup 1 --cpu-baseline off --no-roofline --also off ${BENCH_ARGS:-} > "$OUT/step.log" 2>&1
  echo "step pass rc=$?"; tail -c 400 "$OUT/step.log"; echo
  F=$(find "$OUT/step" -name '*counter_collection.csv' | head -1); K=$(find "$OUT/step" -name '*kernel_trace.csv' | head -1)
  [ -n "$F" ] && python "$ROOT/tools/pmc_agg.py" "$F" "$OUT/step_pmc_by_kernel.csv" $K
  rm -rf "$OUT/step"
fi
SHAPES=("full g4.first" "full g3.first" "full up_g4.last" "full up_g4.first" "full up_g3.first" "gru3.h-step B96" "gru4.h-step B96" "gru2.h-step B96")
i=0
for CTRS in "$SETA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  CONV_BENCH_SKIP_OLD=1 CONV_BENCH_ITERS=3 timeout 400 rocprofv3 --kernel-trace --pmc $CTRS -f csv -d "$OUT/m$i" -o pmc -- \
      python "$ROOT/tools/conv_bench.py" --prec=bf16x3 --bwd --groups=96 --phases-only ${CONV_BENCH_EXTRA:-} "${SHAPES[@]}" > "$OUT/modes_pass$i.log" 2>&1
  echo "modes pass $i rc=$?"
  F=$(find "$OUT/m$i" -name '*counter_collection.csv' | head -1); K=$(find "$OUT/m$i" -name '*kernel_trace.csv' | head -1)
  [ -n "$F" ] && python "$ROOT/tools/pmc_agg.py" "$F" "$OUT/modes_pmc_pass$i.csv" $K conv wgrad
  rm -rf "$OUT/m$i"
done
cat "$OUT/modes_pass1.log" | cut -c1-250
python "$ROOT/tools/pmc_classes.py" "$OUT" "$OUT/pmc_classes.json" | tail -40
