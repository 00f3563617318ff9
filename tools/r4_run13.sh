#!/bin/bash
# round 4, pass 13: the prefetch test again; the three-step golden in `mixed` with the prefetch on and off (is its marginal
# sampler.conv_1x1.bias entry the prefetch or run-to-run noise?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run13}
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python -m pytest tests/test_gpu_sn_prefetch.py -m gpu -q > "$OUT/pytest_prefetch.log" 2>&1; echo "prefetch test rc=$?"; grep -E "^FAILED|passed|failed|^E  " "$OUT/pytest_prefetch.log" | cut -c1-250 | head -12
for pf in 1 0 1 0; do
  DGMR_SN_PREFETCH=$pf timeout 300 python -m pytest "tests/test_training_steps_adv.py::test_hip_training_steps_adv_match_reference" -m gpu -q > "$OUT/pytest_adv_pf$pf.log" 2>&1; echo "adv prefetch=$pf rc=$?"
  grep -E "passed|failed" "$OUT/pytest_adv_pf$pf.log" | tail -1; grep -h "generator.sampler.conv_1x1.bias  " "$OUT/pytest_adv_pf$pf.log" gpurun_out/band_tables.log 2>/dev/null | tail -4 | cut -c1-200
done
