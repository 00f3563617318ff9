#!/bin/bash
# round 4, pass 14: the streaming 1x1-conv kernel and the prefetch tests, then a short bench line with the kernel on and off
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run14}
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python -m pytest tests/test_gpu_sn_prefetch.py tests/test_gpu_kernels.py -m gpu -q -k "prefetch or streaming_1x1 or four_channel" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest_gpu.log" | cut -c1-250 | tail; grep -E "^E  " "$OUT/pytest_gpu.log" | cut -c1-250 | head -20
for on in 1 0; do
  DGMR_CONV1X1=$on timeout 300 python bench.py --steps 8 --warmup 4 --also off --cpu-baseline off > "$OUT/bench_1x1_$on.json" 2>"$OUT/bench_1x1_$on.err"; echo "bench rc=$?"
  python - <<P
import json
d=json.loads(open("$OUT/bench_1x1_$on.json").read().strip().splitlines()[-1]); print("conv1x1=$on ms/step", d["ms_per_step"], d["step_ms"])
for r in d["roofline"]["per_kernel_detail"]:
    if "1x1" in r["kernel"] and r["total_ms"] > 0.5: print("  %-78s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
P
done
