#!/bin/bash
# rocprofv3 kernel trace of one steady-state step; keeps the trace rows of the last step (small) for offline analysis
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4_prof}
shift || true
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof" -o trace -- python "$ROOT/bench.py" --steps 1 --warmup 2 --cpu-baseline off --no-roofline --also off "$@" > "$OUT/prof_run.log" 2>&1
echo "rocprof rc=$?"
cd "$ROOT"
KT=$(find "$OUT/prof" -name '*kernel_trace.csv' | head -1)
LAST_MS=$(python -c "import json,sys; print(1.02*json.loads([l for l in open('$OUT/prof_run.log') if l.startswith('{')][-1])['ms_per_step'])" 2>/dev/null || echo 1000)
echo "window $LAST_MS ms"
python - "$KT" "$OUT/last_step_trace.csv" "$LAST_MS" <<'P'
import csv, sys
src, dst, last_ms = sys.argv[1], sys.argv[2], float(sys.argv[3])
rows = list(csv.DictReader(open(src)))
end = max(int(r["End_Timestamp"]) for r in rows)
cut = end - int(last_ms * 1e6)
keep = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
cols = [c for c in ("Kernel_Name", "Queue_Id", "Stream_Id", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count") if c in rows[0]]
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(cols)
    for r in keep:
        w.writerow([r[c][:120] if c == "Kernel_Name" else r[c] for c in cols])
print(len(keep), "rows kept; columns:", list(rows[0].keys()))
P
python tools/trace_by_grid.py "$KT" "$OUT/kernel_by_grid.csv" $LAST_MS "$OUT/kernel_stats_last_step.csv"
python tools/trace_gaps.py "$KT" "$OUT/kernel_gaps.txt" 20 $LAST_MS
rm -rf "$OUT/prof"
ls -la "$OUT"
