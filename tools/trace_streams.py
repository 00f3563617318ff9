"""Per-queue view of a rocprofv3 kernel-trace CSV (run on the GPU box):  python tools/trace_streams.py <kernel_trace.csv> <out.txt> [last_ms]

Which stream is the critical path of a training step, what is on it, and what a co-running stream costs it:
  * per queue: launches, busy time (union of its kernel intervals), share of the window, its biggest kernels;
  * concurrency: time with 0 / 1 / 2 / 3+ queues active;
  * for the busiest queue (the main chain): every kernel's duration split by whether ANOTHER queue had a kernel running during at least half
    of it - the same kernel class alone vs beside the weight-gradient stream (what the side stream costs the chain);
  * the main queue's time by kernel family (window convs / other convs / everything else).
last_ms > 0: only kernels that start in the final last_ms milliseconds (the steady-state step of `bench.py --warmup 2 --steps 1`)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name[:78]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    ev = []
    with open(src) as f:
        for r in csv.DictReader(f):
            q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"), q))
    skip = float(__import__('os').environ.get('TRACE_SKIP_TAIL_MS', '0'))  # drop the trace's last ms (post-loop diagnostics)
    if skip > 0:
        hi = max(e for _, e, _, _ in ev) - int(skip * 1e6)
        ev = [x for x in ev if x[0] < hi]
    if last_ms > 0:
        cut = max(e for _, e, _, _ in ev) - int(last_ms * 1e6)
        ev = [x for x in ev if x[0] >= cut]
    ev.sort()
    t0, t1 = min(s for s, _, _, _ in ev), max(e for _, e, _, _ in ev)
    span = (t1 - t0) / 1e6
    out = [f"window {span:.1f} ms, {len(ev)} launches"]
    byq = defaultdict(list)
    for s, e, n, q in ev:
        byq[q].append((s, e, n))

    def union(iv):
        tot, cs, ce = 0, None, None
        for s, e in sorted(iv):
            if cs is None or s > ce:
                if cs is not None:
                    tot += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        return tot + ((ce - cs) if cs is not None else 0)

    busy = {q: union([(s, e) for s, e, _ in v]) for q, v in byq.items()}
    order = sorted(byq, key=lambda q: -busy[q])
    for q in order:
        v = byq[q]
        ks = defaultdict(lambda: [0, 0])
        for s, e, n in v:
            ks[short(n)][0] += 1
            ks[short(n)][1] += e - s
        out.append(f"\nqueue {q}: {len(v)} launches, busy {busy[q]/1e6:.1f} ms = {100*busy[q]/(t1-t0):.1f} % of the window, kernel time {sum(e-s for s,e,_ in v)/1e6:.1f} ms")
        for k, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:14]:
            out.append(f"    {t/1e6:8.2f} ms {c:5d}  {k}")
    # concurrency histogram: sweep over interval edges of per-queue busy unions
    edges = []
    for q, v in byq.items():
        cs = ce = None
        for s, e, _ in sorted(v):
            if cs is None or s > ce:
                if cs is not None:
                    edges += [(cs, 1), (ce, -1)]
                cs, ce = s, e
            else:
                ce = max(ce, e)
        if cs is not None:
            edges += [(cs, 1), (ce, -1)]
    edges.sort()
    hist = defaultdict(int)
    lvl, prev = 0, t0
    for t, d in edges:
        hist[min(lvl, 3)] += t - prev
        lvl += d
        prev = t
    out.append("\nqueues active at once: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {hist[k]/1e6:.1f} ms" for k in sorted(hist)))
    # the main queue's kernels alone vs beside another queue
    main_q = order[0]
    others = sorted((s, e) for q in order[1:] for s, e, _ in byq[q])
    import bisect

    starts = [s for s, _ in others]
    # prefix maximum of ends for overlap queries
    cover = []
    cs = ce = None
    for s, e in others:
        if cs is None or s > ce:
            if cs is not None:
                cover.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        cover.append((cs, ce))
    cstarts = [s for s, _ in cover]

    def overlap(s, e):
        i = max(0, bisect.bisect_right(cstarts, s) - 1)
        tot = 0
        while i < len(cover) and cover[i][0] < e:
            tot += max(0, min(e, cover[i][1]) - max(s, cover[i][0]))
            i += 1
        return tot

    cls = defaultdict(lambda: [0, 0, 0, 0])  # alone n, alone ns, beside n, beside ns
    for s, e, n in byq[main_q]:
        ov = overlap(s, e)
        c = cls[short(n)]
        if ov * 2 >= (e - s):
            c[2] += 1
            c[3] += e - s
        else:
            c[0] += 1
            c[1] += e - s
    out.append(f"\nmain queue {main_q}: kernels alone vs with another queue active for at least half of their duration (count, total ms, average us)")
    tot_alone = sum(c[1] for c in cls.values())
    tot_beside = sum(c[3] for c in cls.values())
    out.append(f"    total alone {tot_alone/1e6:.1f} ms, beside {tot_beside/1e6:.1f} ms")
    for k, c in sorted(cls.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:30]:
        a = f"{c[0]:5d} {c[1]/1e6:8.2f} ms {c[1]/max(c[0],1)/1e3:9.1f} us" if c[0] else " " * 33
        b = f"{c[2]:5d} {c[3]/1e6:8.2f} ms {c[3]/max(c[2],1)/1e3:9.1f} us" if c[2] else ""
        out.append(f"    {k:78s} alone {a} | beside {b}")
    # the main queue idle while another queue works: exposed tails of the side streams (joins before an optimiser step) and waits on them
    mk = sorted(byq[main_q])
    tails = []
    for (s0, e0, n0), (s1, e1, n1) in zip(mk, mk[1:]):
        if s1 > e0:
            ov = overlap(e0, s1)
            if ov > 0:
                tails.append((ov, (s1 - e0), e0 - t0, short(n0), short(n1)))
    tot_t = sum(t[0] for t in tails)
    out.append(f"\nmain queue {main_q} idle while another queue is busy: {tot_t/1e6:.1f} ms in {len(tails)} gaps; the largest (other-queue busy ms | gap ms | at ms | main kernel before -> after):")
    for ov, gap, at, n0, n1 in sorted(tails, reverse=True)[:14]:
        out.append(f"    {ov/1e6:7.2f} | {gap/1e6:7.2f} | {at/1e6:7.1f} | {n0[:50]} -> {n1[:50]}")
    fam = defaultdict(int)
    for s, e, n in byq[main_q]:
        sn = short(n)
        f_ = "window conv fwd/dgrad" if "conv3x3_glds" in sn or "conv3x3_win" in sn or "conv3x3_ws" in sn else (
            "weight gradient" if "wgrad" in sn else ("other conv (implicit GEMM, 1x1, stem)" if sn.startswith("conv") else (
                "spectral norm" if sn.startswith("sn_") else ("BatchNorm" if sn.startswith("bn_") else ("torch (at::native / rocclr)" if "at::" in sn or "rocclr" in sn else "other library kernels")))))
        fam[f_] += e - s
    out.append(f"\nmain queue {main_q} by family: " + "; ".join(f"{k} {v/1e6:.1f} ms" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:60] + out[-40:]))


if __name__ == "__main__":
    main()
