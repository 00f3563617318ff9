#!/bin/bash
# round 4, pass 10: every GPU test except the nine-minute full-step case (no -x: all failures at once)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_run10}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullstep.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest_gpu.log" | cut -c1-250 | tail -40
