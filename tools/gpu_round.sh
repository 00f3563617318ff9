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
      python "$ROOT/bench.py" "$@" --steps 1 --warmup 2 --cpu-baseline off --no-roofline --also off > "$OUT/prof_run.log" 2>&1
  echo "rocprof rc=$?"
  cd "$ROOT"
  # keep only the summaries (the raw per-dispatch trace / sqlite db can be hundreds of MB; gpurun_out is capped at 64 MiB)
  mkdir -p "$OUT/prof_keep"
  find "$OUT/prof" -name '*stats*.csv' -exec cp {} "$OUT/prof_keep/" \;
  KT=$(find "$OUT/prof" -name '*kernel_trace.csv' | head -1)
  # steady state = the last step of the run: window = its ms_per_step as bench.py reported it under the profiler (+2 %)
  LAST_MS=${GAPS_LAST_MS:-$(python -c "import json,sys; print(1.02*json.loads([l for l in open('$OUT/prof_run.log') if l.startswith('{')][-1])['ms_per_step'])" 2>/dev/null || echo 0)}
  echo "steady-state window: $LAST_MS ms"
  [ -n "$KT" ] && python tools/trace_by_grid.py "$KT" "$OUT/prof_keep/kernel_by_grid.csv" $LAST_MS "$OUT/prof_keep/kernel_stats_last_step.csv" && head -3 "$KT" > "$OUT/prof_keep/kernel_trace_head.csv"
  [ -n "$KT" ] && python tools/trace_gaps.py "$KT" "$OUT/prof_keep/kernel_gaps.txt" 20 $LAST_MS
  rm -rf "$OUT/prof"
  ls -la "$OUT/prof_keep"
fi

if [ "${SKIP_CONVBENCH:-0}" != "1" ]; then
  timeout 600 python tools/conv_bench.py --bwd > "$OUT/conv_bench.log" 2>&1
  cat "$OUT/conv_bench.log"
fi
