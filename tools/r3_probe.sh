#!/bin/bash
# Round-3 kernel probes (GPU box): launch-size sweep of the dominant layer, phase split of the window kernel (staging / matrix loop /
# epilogue, dgmr_debug_flags), weight-gradient kernels at the full draw batch.  Output: gpurun_out/<tag>/probe_*.log
TAG=${1:-r3p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
{
  echo "== launch-size sweep, up_g4.first (phase path), bf16x3"
  timeout 300 python tools/conv_bench.py --prec=bf16x3 --phases-only "up_g4.first T18" "nsweep up_g4.first" "full up_g4.first"
} > "$OUT/probe_nsweep.log" 2>&1
for dbg in 0 1 2 3; do
  {
    echo "== window kernel phases, dbg=$dbg (1: no epilogue, 2: one halo only, 3: both), bf16x3"
    timeout 300 python tools/conv_bench.py --prec=bf16x3 --phases-only --dbg=$dbg "full g4.first" "full up_g3.last" "full g3.first" "full g2.first" \
        "full up_g4.last" "half up_g4.last" "full up_g4.first" "full up_g3.first" "gru4.h-step B96" "gru3.h-step B96" "gru1.h-step B96" "tempD.d1.last"
  } > "$OUT/probe_dbg$dbg.log" 2>&1
done
{
  echo "== weight gradients at the full draw batch (108 call groups), bf16x3"
  timeout 600 python tools/conv_bench.py --prec=bf16x3 --bwd --groups=108 "full g4.first" "full up_g3.last" "full g3.first" "full g2.first" "full up_g4.last" \
      "full up_g4.first" "full up_g3.first"
} > "$OUT/probe_wgrad.log" 2>&1
tail -n +1 "$OUT"/probe_*.log
