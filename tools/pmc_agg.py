"""Aggregate a rocprofv3 `--pmc` counter_collection CSV (one row per dispatch x counter) by (kernel, grid): dispatch count, counter SUMS,
and - when the kernel-trace CSV of the same run is given - the summed durations (joined on Dispatch_Id).  Run on the GPU box.

    python tools/pmc_agg.py <counter_collection.csv> <out.csv> [kernel_trace.csv] [name filter ...]

Sums, not means: a time-weighted busy share over a set of launches is sum(busy) / sum(available cycles).
"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
trace = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3].endswith(".csv") else None
filters = [a for a in sys.argv[3:] if not a.endswith(".csv")] or ["conv", "wgrad", "bn_", "sn_", "pool", "gru", "head", "adam", "axpby", "split"]
dur = {}
if trace:
    with open(trace) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
first = {}
counters = []
with open(src) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "")
        if not any(s in name for s in filters):
            continue
        key = (name[:150], r.get("Grid_Size", ""))
        c = r.get("Counter_Name", "")
        if c not in counters:
            counters.append(c)
        agg[key][c] += float(r.get("Counter_Value", 0) or 0)
        did = r.get("Dispatch_Id", "")
        if key not in first:
            first[key] = int(did or 0)
        if did not in seen[key]:
            seen[key].add(did)
            agg[key]["__ns"] += dur.get(did, 0)
with open(dst, "w") as f:
    f.write("kernel,grid,dispatches,first_dispatch,total_ms," + ",".join(counters) + "\n")
    for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", kv[1]["__ns"])):
        f.write(f"\"{key[0]}\",{key[1]},{len(seen[key])},{first[key]},{c['__ns'] / 1e6:.3f}," + ",".join(f"{c.get(k, 0.0):.0f}" for k in counters) + "\n")
