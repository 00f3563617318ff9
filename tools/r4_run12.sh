#!/bin/bash
# round 4, pass 12: spectral-norm prefetch - its own test, the training-step goldens, a short bench line with and without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run12}
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_sn_prefetch.py tests/test_gpu_draws.py tests/test_training_step.py tests/test_training_steps_adv.py tests/test_gpu_ddp.py -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest_gpu.log" | cut -c1-250 | tail; grep -E "^E  " "$OUT/pytest_gpu.log" | cut -c1-250 | head -20
for pf in 1 0; do
  DGMR_SN_PREFETCH=$pf timeout 300 python bench.py --steps 8 --warmup 4 --also off --cpu-baseline off > "$OUT/bench_pf$pf.json" 2>"$OUT/bench_pf$pf.err"; echo "bench rc=$?"
  python - <<P
import json
d=json.loads(open("$OUT/bench_pf$pf.json").read().strip().splitlines()[-1]); print("prefetch=$pf ms/step", d["ms_per_step"], d["step_ms"])
P
done
