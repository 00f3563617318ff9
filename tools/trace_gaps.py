"""Idle-time analysis of a rocprofv3 kernel-trace CSV (run on the GPU box): where does the device wait for the host?

    python tools/trace_gaps.py <kernel_trace.csv> <out.txt> [min_gap_us=20] [last_ms=0]

last_ms > 0 restricts the analysis to the final last_ms milliseconds of the trace (the steady-state step).

Kernels are sorted by start time; a gap is the time between the latest end seen so far and the next start.  Reports the busy /
idle split of the last step in the trace (the span after the largest gap is NOT special-cased: run with --warmup 1 --steps 1
and read the totals as "two steps"), the gap histogram, idle time attributed to the kernel that FOLLOWS the gap, and the
largest gaps with their neighbours.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    return name.split("(")[0][:70]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    ev = []
    queues = []
    with open(src) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
            queues.append((ev[-1][0], ev[-1][1], ev[-1][2], r.get("Queue_Id") or r.get("Stream_Id") or "?"))
    ev.sort()
    last_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    skip = float(__import__('os').environ.get('TRACE_SKIP_TAIL_MS', '0'))  # drop the trace's last ms (post-loop diagnostics)
    if skip > 0:
        hi = max(e for _, e, _ in ev) - int(skip * 1e6)
        ev = [x for x in ev if x[0] < hi]
        queues = [q for q in queues if q[0] < hi]
    if last_ms > 0:
        cut = max(e for _, e, _ in ev) - int(last_ms * 1e6)
        ev = [x for x in ev if x[0] >= cut]
    t_end = ev[0][0]
    prev = None
    gaps = []
    busy = 0
    for s, e, n in ev:
        if s > t_end:
            gaps.append((s - t_end, prev, n, s))
            busy += e - s
        else:
            busy += max(0, e - max(s, t_end))
        if e > t_end:
            t_end, prev = e, n
    span = t_end - ev[0][0]
    out = []
    out.append(f"kernels {len(ev)}  span {span/1e6:.1f} ms  busy {busy/1e6:.1f} ms  idle {(span-busy)/1e6:.1f} ms ({100*(span-busy)/span:.1f} %)")
    edges = [0, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 5000, 1e9]
    hist = [[0, 0.0] for _ in edges]
    for g, *_ in gaps:
        us = g / 1e3
        for i in range(len(edges) - 1):
            if edges[i] <= us < edges[i + 1]:
                hist[i][0] += 1
                hist[i][1] += us
                break
    out.append("gap histogram (us): count, total ms")
    for i in range(len(edges) - 1):
        out.append(f"  [{edges[i]:>6g}, {edges[i+1]:>6g})  {hist[i][0]:7d}  {hist[i][1]/1e3:9.2f}")
    by_next = defaultdict(lambda: [0, 0.0])
    for g, p, n, _ in gaps:
        a = by_next[short(n)]
        a[0] += 1
        a[1] += g / 1e3
    out.append("idle attributed to the kernel after the gap (top 25): count, total ms, avg us")
    for k, (c, t) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:25]:
        out.append(f"  {k:70s} {c:7d} {t/1e3:9.2f} {t/c:8.1f}")
    by_prev = defaultdict(lambda: [0, 0.0])
    for g, p, n, _ in gaps:
        a = by_prev[short(p or "")]
        a[0] += 1
        a[1] += g / 1e3
    out.append("idle attributed to the kernel before the gap (top 15): count, total ms, avg us")
    for k, (c, t) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:15]:
        out.append(f"  {k:70s} {c:7d} {t/1e3:9.2f} {t/c:8.1f}")
    out.append(f"largest gaps (>= {min_gap} us, top 40): gap us | at ms | before -> after")
    t0 = ev[0][0]
    for g, p, n, s in sorted(gaps, key=lambda x: -x[0])[:40]:
        if g / 1e3 < min_gap:
            break
        out.append(f"  {g/1e3:9.1f} | {(s-t0)/1e6:9.1f} | {short(p or '')} -> {short(n)}")
    # what ran around the five largest gaps: queue | start ms | duration us | kernel (the 8 launches on either side, every queue)
    queues.sort()
    starts = [q[0] for q in queues]
    import bisect

    for g, p, n, s in sorted(gaps, key=lambda x: -x[0])[:5]:
        if g / 1e3 < 200:
            break
        i = bisect.bisect_left(starts, s)
        out.append(f"around the {g/1e3:.0f} us gap at {(s-t0)/1e6:.1f} ms:")
        for qs, qe, qn, qq in queues[max(0, i - 8):i + 8]:
            out.append(f"    q{qq:>3s} | {(qs-t0)/1e6:9.3f} | {(qe-qs)/1e3:9.1f} | {short(qn)}")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:16]))


if __name__ == "__main__":
    main()
