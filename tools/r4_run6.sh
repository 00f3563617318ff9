#!/bin/bash
# stagger experiment: the one-role window kernel ("lib" column) with the second resident set of workgroups started late
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run6}
mkdir -p "$OUT"
cd "$ROOT"
for d in 0 256 512; do
  for pct in 100 200; do
    [ $d = 0 ] && [ $pct = 200 ] && continue
    DGMR_STAGGER_PCT=$pct timeout 300 python tools/ws_check.py --big-only --dbg=$d "full g4.first" "full g4.last" "full g4 dgrad" "full up_g3.last" "full g3.first" "full up_g4.first" "full up_g4 dgrad" "full g2.first" "B16 g4" > "$OUT/stag_${d}_$pct.log" 2>&1
    echo "== dbg=$d pct=$pct"; sed 's/.*| ref6.*lib/lib/' "$OUT/stag_${d}_$pct.log" | grep -v amdgpu.ids
  done
done
