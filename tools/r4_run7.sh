#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run7}
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest_kernels.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_kernels.log"
for d in 0 3; do
  timeout 300 python tools/ws_check.py --big-only --dbg=$d "full g4.first" "full g4.last" "full up_g4.first" "full up_g4 dgrad" "full g2.first" "B16 g4" > "$OUT/ws_dbg$d.log" 2>&1
  echo "== dbg=$d"; sed 's/.*| y==ref6: \([A-Za-z]*\).*| ref6/\1 ref6/' "$OUT/ws_dbg$d.log" | grep -v amdgpu.ids
done
timeout 200 python tools/conv_bench.py --prec=bf16x3 "full up_g4.last" "half up_g4.last" "real up_g4.last" "gru4.h-step B96" "c64 T18" 2>&1 | grep -v amdgpu | cut -c1-120
timeout 300 python bench.py --steps 5 --warmup 2 --also off --cpu-baseline off > "$OUT/bench.json" 2>/dev/null
python - <<P
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("ms/step", d["ms_per_step"], d["step_ms"])
for r in d["roofline"]["per_kernel_detail"][:16]: print("  %-70s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
P
