#!/bin/bash
# round 4, pass 16 (last): prefetch and phase-wgrad tests on the last tree, then the full bench line (all legs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run16}
mkdir -p "$OUT"
cd "$ROOT"
timeout 120 python -m pytest tests/test_gpu_sn_prefetch.py tests/test_gpu_kernels.py -m gpu -q -k "prefetch or by_phases or weight_gradient_paths" > "$OUT/pytest.log" 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed|^E  " "$OUT/pytest.log" | cut -c1-220 | head -8
timeout 200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2>"$OUT/bench.err"; echo "bench rc=$?"
python - <<P
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("ms/step", d["ms_per_step"], d["value"], d["step_ms"][:5])
for k, v in d.get("also", {}).items(): print(k, v.get("ms_per_step"))
print(d.get("cpu_baseline", {}).get("seconds_per_step"))
P
