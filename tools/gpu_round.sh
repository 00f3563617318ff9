#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel stats, conv micro-benchmark.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [bench args...]
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > "$OUT/device.txt"
nproc > "$OUT/host_cores.txt"; lscpu | grep -E "Model name|^CPU\(s\)" >> "$OUT/host_cores.txt"

if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu ${PYTEST_FLAGS:--x -q} > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi

if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench rc=$?"
  tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi

if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof" -o trace -- \
      python "$ROOT/bench.py" "$@" --steps 1 --warmup 1 --cpu-baseline off --no-roofline --also off > "$OUT/prof_run.log" 2>&1
  echo "rocprof rc=$?"
  cd "$ROOT"
  # keep only the summaries (the raw per-dispatch trace / sqlite db can be hundreds of MB; gpurun_out is capped at 64 MiB)
  mkdir -p "$OUT/prof_keep"
  find "$OUT/prof" -name '*stats*.csv' -exec cp {} "$OUT/prof_keep/" \;
  KT=$(find "$OUT/prof" -name '*kernel_trace.csv' | head -1)
  [ -n "$KT" ] && python tools/trace_by_grid.py "$KT" "$OUT/prof_keep/kernel_by_grid.csv" && head -3 "$KT" > "$OUT/prof_keep/kernel_trace_head.csv"
  [ -n "$KT" ] && python tools/trace_gaps.py "$KT" "$OUT/prof_keep/kernel_gaps.txt" 20 ${GAPS_LAST_MS:-0}
  rm -rf "$OUT/prof"
  ls -la "$OUT/prof_keep"
fi

if [ "${SKIP_CONVBENCH:-0}" != "1" ]; then
  timeout 600 python tools/conv_bench.py --bwd > "$OUT/conv_bench.log" 2>&1
  cat "$OUT/conv_bench.log"
fi
