#!/bin/bash
# One parameterised GPU-box pass (replaces the per-pass r4_run*.sh scripts).  Through gpurun, from the repo root:
#   bash tools/gpu_call.sh <tag> [steps...]      steps, in the order given:
#     tests:<pytest -k expression or file list>   selected -m gpu tests           -> pytest_<n>.log
#     alltests                                    the whole -m gpu suite          -> pytest_all.log
#     bench:[VAR=v ...] <bench.py args>           one bench line + detail file (leading VAR=v words: environment)    -> bench_<n>.json / bench_<n>_detail.json
#     prof:<bench.py args>                        rocprofv3 --kernel-trace --stats of a 1-step run (summaries only are kept)
#     pmc                                         tools/r5_pmc.sh (counters for every conv class)
#     convbench:<conv_bench.py args>              kernel micro-benchmarks         -> convbench_<n>.log
#     py:<script and args>                        any tools/*.py                  -> py_<n>.log
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-r5}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
n=0
for STEP in "$@"; do
  n=$((n+1))
  KIND=${STEP%%:*}; ARG=""; [ "$STEP" != "$KIND" ] && ARG=${STEP#*:}
  T0=$(date +%s)
  case "$KIND" in
    tests)
      if [[ "$ARG" == tests/* ]]; then SEL="$ARG"; else SEL="tests -k \"$ARG\""; fi
      eval timeout 1200 python -m pytest $SEL -m gpu -q -x > "$OUT/pytest_$n.log" 2>&1; echo "[$n] tests rc=$?"
      grep -E "^FAILED|^ERROR|passed|failed|^E  " "$OUT/pytest_$n.log" | cut -c1-260 | head -12 ;;
    alltests)
      timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_FLAGS:-} > "$OUT/pytest_all.log" 2>&1; echo "[$n] alltests rc=$?"
      grep -E "^FAILED|^ERROR|passed|failed" "$OUT/pytest_all.log" | cut -c1-260 | head -20 ;;
    bench)
      ENVV=""  # leading VAR=value words of the argument go to the environment (A/B switches)
      while [[ "${ARG%% *}" == *=* && "${ARG%% *}" != --* ]]; do ENVV="$ENVV ${ARG%% *}"; ARG="${ARG#* }"; done
      timeout 900 env $ENVV python bench.py $ARG --detail "gpurun_out/$TAG/bench_${n}_detail.json" > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"; echo "[$n] bench rc=$?"
      tail -c 3500 "$OUT/bench_$n.json"; echo; tail -3 "$OUT/bench_$n.err" ;;
    prof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof$n" -o trace -- \
          python "$ROOT/bench.py" $ARG --steps 1 --warmup 2 --cpu-baseline off --no-roofline --also off > "$OUT/prof_run_$n.log" 2>&1; echo "[$n] rocprof rc=$?" )
      mkdir -p "$OUT/prof_keep_$n"
      find "$OUT/prof$n" -name '*stats*.csv' -exec cp {} "$OUT/prof_keep_$n/" \;
      KT=$(find "$OUT/prof$n" -name '*kernel_trace.csv' | head -1)
      LAST_MS=$(python -c "import json; print(1.02*json.loads([l for l in open('$OUT/prof_run_$n.log') if l.startswith('{')][-1])['ms_per_step'])" 2>/dev/null || echo 0)
      echo "steady-state window: $LAST_MS ms"
      [ -n "$KT" ] && python tools/trace_by_grid.py "$KT" "$OUT/prof_keep_$n/kernel_by_grid.csv" $LAST_MS "$OUT/prof_keep_$n/kernel_stats_last_step.csv" \
        && python tools/trace_gaps.py "$KT" "$OUT/prof_keep_$n/kernel_gaps.txt" 20 $LAST_MS \
        && python tools/trace_streams.py "$KT" "$OUT/prof_keep_$n/kernel_streams.txt" $LAST_MS > /dev/null
      rm -rf "$OUT/prof$n"; head -25 "$OUT/prof_keep_$n/kernel_stats_last_step.csv" | cut -c1-160 ;;
    pmc)
      SKIP_STEP=${ARG:-0} bash tools/r5_pmc.sh "$TAG/pmc" ;;
    convbench)
      timeout 600 python tools/conv_bench.py $ARG > "$OUT/convbench_$n.log" 2>&1; echo "[$n] convbench rc=$?"; cut -c1-300 "$OUT/convbench_$n.log" ;;
    py)
      timeout 900 python $ARG > "$OUT/py_$n.log" 2>&1; echo "[$n] py rc=$?"; tail -40 "$OUT/py_$n.log" | cut -c1-300 ;;
    *) echo "unknown step $STEP" ;;
  esac
  echo "[$n] $KIND took $(( $(date +%s) - T0 )) s"
done
