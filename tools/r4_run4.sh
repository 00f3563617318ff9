#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run4}
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python tools/ws_check.py --no-time > "$OUT/ws_check_small.log" 2>&1
echo "small rc=$?"; grep -c " OK" "$OUT/ws_check_small.log"; grep "FAIL\|Error\|error" "$OUT/ws_check_small.log" | head
for d in 0 3; do
  timeout 300 python tools/ws_check.py --big-only --dbg=$d "full g4.first" "full g4.last" "full g4 dgrad" "full up_g3.last" "full up_g4.first" "full up_g4 dgrad" "full g2.first" "B16 g4" > "$OUT/ws_dbg$d.log" 2>&1
  echo "== dbg=$d"; sed 's/.*| y==ref6: \([A-Za-z]*\).*| ref6/\1 ref6/' "$OUT/ws_dbg$d.log" | grep -v amdgpu.ids
done
