"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, total and average duration.  Run on the GPU box.

    python tools/trace_by_grid.py <kernel_trace.csv> <by_grid.csv> [last_ms] [by_name.csv]

last_ms > 0 restricts the aggregation to kernels that START in the final last_ms milliseconds of the trace: with
`bench.py --warmup 2 --steps 1` and last_ms ~ one step, that is the steady-state step alone (the first steps also trace the
spectral-norm plans, split weights for the first time, allocate ...).  by_name.csv: the same window aggregated by kernel name only.
"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
by_name = sys.argv[4] if len(sys.argv) > 4 else None
ev = []
with open(src) as f:
    rd = csv.DictReader(f)
    for r in rd:
        name = r.get("Kernel_Name") or r.get("Name")
        grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid))
_skip = float(__import__('os').environ.get('TRACE_SKIP_TAIL_MS', '0'))  # drop the trace's last ms (bench.py's post-loop diagnostics under a process group)
if _skip > 0:
    _hi = max(e for _, e, _, _ in ev) - int(_skip * 1e6)
    ev = [x for x in ev if x[0] < _hi]
if last_ms > 0:
    cut = max(e for _, e, _, _ in ev) - int(last_ms * 1e6)
    ev = [x for x in ev if x[0] >= cut]
agg = defaultdict(lambda: [0, 0.0])
agn = defaultdict(lambda: [0, 0.0])
for s, e, name, grid in ev:
    for a in (agg[(name, grid)], agn[name]):
        a[0] += 1
        a[1] += e - s
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
with open(dst, "w") as f:
    f.write("kernel,grid,calls,total_ms,avg_us,percent\n")
    for (name, grid), (n, t) in rows[:200]:
        f.write(f"\"{name[:110]}\",{grid},{n},{t/1e6:.3f},{t/n/1e3:.2f},{100*t/tot:.2f}\n")
if by_name:
    span = (max(e for _, e, _, _ in ev) - min(s for s, _, _, _ in ev)) / 1e6
    with open(by_name, "w") as f:
        f.write(f"# window {span:.1f} ms, {len(ev)} kernel launches, kernel time {tot/1e6:.1f} ms\n")
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for name, (n, t) in sorted(agn.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{name[:140]}\",{n},{t/1e6:.3f},{t/n/1e3:.2f},{100*t/tot:.2f}\n")
