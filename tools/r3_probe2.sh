#!/bin/bash
# A/B of the window kernels' 16-byte epilogue (dgmr_debug_flags 8 = lane-per-channel epilogue) on the sampler's layers at the full draw batch
TAG=${1:-r3p2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for dbg in 0 8 1; do
  {
    echo "== dbg=$dbg (0: 16-byte epilogue, 8: lane-per-channel epilogue, 1: no epilogue), bf16x3"
    timeout 300 python tools/conv_bench.py --prec=bf16x3 --phases-only --dbg=$dbg "full g4.first" "full up_g3.last" "full g3.first" "full g2.first" \
        "full up_g4.last" "full up_g4.first" "full up_g3.first" "gru4.h-step B96" "gru3.h-step B96" "gru1.h-step B96" "tempD.d1.last"
  } > "$OUT/probe_epi_dbg$dbg.log" 2>&1
done
tail -n +1 "$OUT"/probe_epi_*.log
