#!/bin/bash
# How long does a finished wave of the window kernel wait for its stores?  (dgmr_debug_flags 64 / 128 / 192: sleep 3.4 / 6.8 / 10.2 us after
# the last store is issued; a launch that does not get slower was waiting at least that long)
TAG=${1:-r3p3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for dbg in 0 64 128 192; do
  {
    echo "== dbg=$dbg, bf16x3"
    timeout 300 python tools/conv_bench.py --prec=bf16x3 --phases-only --dbg=$dbg "full g4.first" "full g3.first" "full up_g4.last" "full up_g4.first" "up_g4.first T18"
  } > "$OUT/probe_tail_dbg$dbg.log" 2>&1
done
tail -n +1 "$OUT"/probe_tail_*.log
