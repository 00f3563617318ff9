"""Aggregate a rocprofv3 counter_collection CSV: per (kernel, counter) mean value per dispatch and dispatch count."""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0, 0.0])
with open(src) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "")
        if "conv" not in name:
            continue
        key = (name[:100], r.get("Counter_Name", ""), r.get("Grid_Size", ""))
        a = agg[key]
        a[0] += 1
        a[1] += float(r.get("Counter_Value", 0) or 0)
with open(dst, "w") as f:
    f.write("kernel,counter,grid,dispatches,mean_value\n")
    for (name, ctr, grid), (n, tot) in sorted(agg.items()):
        f.write(f"\"{name}\",{ctr},{grid},{n},{tot / n:.1f}\n")
