#!/bin/bash
# round 4, pass 9: DBlock restructuring (1x1 shortcut on the pooled map, last conv + pooling as one operator), 16-byte BatchNorm backward:
# every GPU test except the nine-minute full-step case, then a short bench line with the per-kernel rows
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run9}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullstep.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -25 "$OUT/pytest_gpu.log" | cut -c1-300
timeout 400 python bench.py --steps 8 --warmup 3 --also off --cpu-baseline off > "$OUT/bench.json" 2>"$OUT/bench.err"; echo "bench rc=$?"; tail -3 "$OUT/bench.err"
python - <<P
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("ms/step", d["ms_per_step"], d["step_ms"])
for r in d["roofline"]["per_kernel"]: print("  %-50s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
for r in d["roofline"]["per_kernel_detail"][:45]: print("  %-78s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
P
