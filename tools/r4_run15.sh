#!/bin/bash
# round 4, pass 15: weight gradient of the upsampling convs by phases - its kernel test, the goldens with it switched on, bench on / off
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run15}
mkdir -p "$OUT"
cd "$ROOT"
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "by_phases or weight_gradient_paths" > "$OUT/pytest_kernel.log" 2>&1; echo "kernel test rc=$?"; grep -E "^FAILED|passed|failed|^E  " "$OUT/pytest_kernel.log" | cut -c1-220 | head -12
DGMR_WGRAD_PHASES=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_training_step.py -m gpu -q > "$OUT/pytest_goldens_phases_on.log" 2>&1; echo "goldens (phases on) rc=$?"; grep -E "^FAILED|passed|failed" "$OUT/pytest_goldens_phases_on.log" | cut -c1-220 | tail -5
for on in 1 0; do
  DGMR_WGRAD_PHASES=$on timeout 200 python bench.py --steps 6 --warmup 3 --also off --cpu-baseline off > "$OUT/bench_phases_$on.json" 2>"$OUT/bench_phases_$on.err"; echo "bench rc=$?"
  python - <<P
import json
d=json.loads(open("$OUT/bench_phases_$on.json").read().strip().splitlines()[-1]); print("wgrad phases=$on ms/step", d["ms_per_step"], d["step_ms"])
for r in d["roofline"]["per_kernel_detail"]:
    if "wgrad" in r["kernel"] and r["total_ms"] > 2: print("  %-78s n=%4d %8.2f ms %7.1f TF" % (r["kernel"], r["launches"], r["total_ms"], r["tflops"]))
P
done
