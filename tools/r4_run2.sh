#!/bin/bash
# Round-4 GPU pass 2: where does the wave-specialised kernel lose time?  Loader work switched off piece by piece.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run2}
mkdir -p "$OUT"
cd "$ROOT"
for d in 0 1 2 3; do
  timeout 300 python tools/ws_check.py --big-only --dbg=$d "full g4.first" "full g4.last" "full up_g4.first" "full up_g4 dgrad" "full g2.first" > "$OUT/ws_dbg$d.log" 2>&1
  echo "== dbg=$d"; sed 's/.*| ref6/ref6/' "$OUT/ws_dbg$d.log" | grep -v amdgpu.ids
done
