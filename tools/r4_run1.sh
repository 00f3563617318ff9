#!/bin/bash
# Round-4 GPU pass 1: wave-specialised window kernel - correctness against the one-role reference kernel, isolated timing, step A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run1}
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
nproc > "$OUT/host.txt"
timeout 400 python tools/ws_check.py > "$OUT/ws_check_small.log" 2>&1
echo "ws_check small rc=$?"; tail -20 "$OUT/ws_check_small.log"
if grep -q "ALL OK" "$OUT/ws_check_small.log"; then
  timeout 400 python tools/ws_check.py --big-only > "$OUT/ws_check_big.log" 2>&1
  echo "ws_check big rc=$?"; cat "$OUT/ws_check_big.log" | tail -20
  timeout 200 python tools/ws_check.py --prec=bf16 --no-time > "$OUT/ws_check_bf16.log" 2>&1
  echo "ws_check bf16 rc=$?"; tail -3 "$OUT/ws_check_bf16.log"
fi
DGMR_WS_AUTO=0 timeout 400 python bench.py --steps 4 --warmup 2 --also off --cpu-baseline off > "$OUT/bench_ws0.json" 2> "$OUT/bench_ws0.err"
echo "bench ws0 rc=$?"; python - <<P
import json
try:
    d=json.loads(open("$OUT/bench_ws0.json").read().strip().splitlines()[-1]); print("ws0 ms/step", d["ms_per_step"], d["step_ms"])
    for r in d["roofline"]["per_kernel_detail"][:40]: print("  %-70s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
except Exception as e: print("parse failed", e)
P
if grep -q "ALL OK" "$OUT/ws_check_small.log"; then
  DGMR_WS_AUTO=1 timeout 400 python bench.py --steps 4 --warmup 2 --also off --cpu-baseline off > "$OUT/bench_ws1.json" 2> "$OUT/bench_ws1.err"
  echo "bench ws1 rc=$?"; tail -3 "$OUT/bench_ws1.err"; python - <<P
import json
try:
    d=json.loads(open("$OUT/bench_ws1.json").read().strip().splitlines()[-1]); print("ws1 ms/step", d["ms_per_step"], d["step_ms"])
    for r in d["roofline"]["per_kernel_detail"][:40]: print("  %-70s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
except Exception as e: print("parse failed", e)
P
fi
