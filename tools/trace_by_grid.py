"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, total and average duration.  Run on the GPU box."""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0, 0.0])
with open(src) as f:
    rd = csv.DictReader(f)
    for r in rd:
        name = r.get("Kernel_Name") or r.get("Name")
        grid = "x".join(str(r.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = agg[(name, grid)]
        a[0] += 1
        a[1] += dur
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
with open(dst, "w") as f:
    f.write("kernel,grid,calls,total_ms,avg_us,percent\n")
    for (name, grid), (n, t) in rows[:200]:
        f.write(f"\"{name[:110]}\",{grid},{n},{t/1e6:.3f},{t/n/1e3:.2f},{100*t/tot:.2f}\n")
