"""profiles/rNN_pmc_classes.json from the passes of tools/r5_pmc.sh (counters for every conv class, and ONE number -
the time-weighted matrix-pipe busy share over all 3x3 conv launches of the real step - against the north star's ">= 40 % MFMA
utilisation for the 3x3 convs").

    python tools/pmc_classes.py <dir with step_pmc_by_kernel.csv, modes_pmc_pass{1,2,3}.csv> <out.json>

  busy share of a set of launches = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8 XCDs)
  (MI355X_MICROARCH.md; GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, so /8 = cycles of the launch, x 1024 = SIMD cycles)
  HBM read = FETCH_SIZE [KB] x 1024 x 2 (gfx950 counts a 128-byte request as 64 bytes), HBM write = WRITE_SIZE [KB] x 1024.
"""
import csv
import json
import os
import re
import sys


def load(path):
    if not os.path.exists(path):
        return []
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            for k, v in list(r.items()):
                if k not in ("kernel", "grid"):
                    r[k] = float(v or 0)
            rows.append(r)
    return rows


def short(name):
    m = re.search(r"(conv\w*_kernel|conv\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else name.split("(")[0][-60:]


def family(name):
    if any(k in name for k in ("conv3x3_glds_kernel", "conv3x3_ws_kernel", "conv3x3_win_kernel")):
        return "3x3 window forward / data gradient"
    if any(k in name for k in ("conv_wgrad_ws_kernel", "conv_wgrad_win_kernel")):
        return "3x3 window weight gradient"
    if "wgrad" in name and not any(k in name for k in ("reduce", "finalize", "sums", "finish")):
        return "im2col weight gradient (1x1, small maps)"
    if any(k in name for k in ("conv_bf16_kernel", "conv_igemm_kernel")):
        return "implicit-GEMM conv (strided, small recurrent steps, 1x1)"
    if "conv1x1" in name:
        return "streaming 1x1 conv"
    if "stem4" in name:
        return "four-channel first conv (exact fp32 MFMA)"
    return None


def derive(r):
    gui = r.get("GRBM_GUI_ACTIVE", 0.0)
    out = {"kernel": short(r["kernel"]), "grid": r["grid"], "dispatches": int(r["dispatches"]), "total_ms": round(r.get("total_ms", 0.0), 3)}
    if gui > 0:
        out["mfma_busy"] = round(r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * gui), 4)
    if r.get("SQ_INSTS_MFMA", 0) > 0:
        out["valu_per_mfma"] = round(r.get("SQ_INSTS_VALU", 0.0) / r["SQ_INSTS_MFMA"], 2)
    if r.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        out["lds_bank_conflict_frac"] = round(r.get("SQ_LDS_BANK_CONFLICT", 0.0) / r["SQ_LDS_IDX_ACTIVE"], 3)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    step = load(os.path.join(src, "step_pmc_by_kernel.csv"))
    out = {"commit": os.environ.get("DGMR_COMMIT", ""),  # (the tree the counters were taken on: the GPU box has no .git - pass it in)
           "steps_under_counters": "2 (bench.py --warmup 1 --steps 1: rows and shares cover both; they do the same work)",
           "definition": "mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8 XCDs) over the launches of a row: "
                         "the share of SIMD cycles in which the matrix pipe was busy, weighted by launch duration (rocprofv3 --pmc over one "
                         "whole training step of bench.py, paper config, per-GPU batch 16, `mixed`; dispatches run serialised under the "
                         "counter pass, so co-running weight-gradient launches do not dilute each other)"}
    fams = {}
    # shares by kernel-trace DURATION: GRBM_GUI_ACTIVE of tiny launches also counts dispatch overhead under the counter pass
    tot_gui = sum(r.get("total_ms", 0.0) for r in step)
    rows = []
    for r in step:
        fam = family(r["kernel"])
        if fam is None:
            continue
        f = fams.setdefault(fam, {"busy": 0.0, "gui": 0.0, "valu": 0.0, "mfma": 0.0, "dispatches": 0, "ms": 0.0})
        f["busy"] += r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f["gui"] += r.get("GRBM_GUI_ACTIVE", 0.0)
        f["valu"] += r.get("SQ_INSTS_VALU", 0.0)
        f["mfma"] += r.get("SQ_INSTS_MFMA", 0.0)
        f["dispatches"] += int(r["dispatches"])
        f["ms"] += r.get("total_ms", 0.0)
        d = derive(r)
        d["family"] = fam
        d["share_of_gpu_cycles"] = round(r.get("total_ms", 0.0) / tot_gui, 4) if tot_gui else None
        rows.append(d)
    if step:
        win = [f for k, f in fams.items() if k.startswith("3x3 window")]
        b, g = sum(f["busy"] for f in win), sum(f["gui"] for f in win)
        out["mfma_util_weighted"] = round(b / (128.0 * g), 4) if g else None
        out["mfma_util_weighted_note"] = "over every launch of the 3x3 window kernels (forward, data gradient, weight gradient) of one step"
        fw = fams.get("3x3 window forward / data gradient")
        if fw and fw["gui"]:
            out["mfma_util_weighted_fwd_dgrad_only"] = round(fw["busy"] / (128.0 * fw["gui"]), 4)
        out["families"] = {k: {"mfma_busy": round(f["busy"] / (128.0 * f["gui"]), 4) if f["gui"] else None,
                               "valu_per_mfma": round(f["valu"] / f["mfma"], 2) if f["mfma"] else None, "dispatches": f["dispatches"],
                               "share_of_gpu_cycles": round(f["ms"] / tot_gui, 4), "total_ms_under_counters": round(f["ms"], 1)}
                           for k, f in sorted(fams.items(), key=lambda kv: -kv[1]["gui"])}
        rows.sort(key=lambda d: -(d["share_of_gpu_cycles"] or 0))
        out["rows"] = [{k: d[k] for k in ("kernel", "grid", "dispatches", "total_ms", "mfma_busy", "valu_per_mfma", "share_of_gpu_cycles") if k in d}
                       for d in rows[:24]]
        out["rows_below_40_percent"] = [f'{d["kernel"]} grid {d["grid"]} ({d["mfma_busy"]:.0%}, {d["share_of_gpu_cycles"]:.1%} of the step)'
                                        for d in rows if d.get("mfma_busy") is not None and d["mfma_busy"] < 0.40 and d["family"].startswith("3x3 window")
                                        and (d["share_of_gpu_cycles"] or 0) >= 0.005]
    # one launch per mode (tools/conv_bench.py): counters + HBM traffic
    m1, m2, m3 = (load(os.path.join(src, f"modes_pmc_pass{i}.csv")) for i in (1, 2, 3))
    fetch = {(r["kernel"], r["grid"]): r for r in m2}
    write = {(r["kernel"], r["grid"]): r for r in m3}
    modes = []
    for r in sorted(m1, key=lambda r: r.get("first_dispatch", 0)):
        if family(r["kernel"]) is None:
            continue
        d = derive(r)
        d["first_dispatch"] = int(r.get("first_dispatch", 0))
        n = max(int(r["dispatches"]), 1)
        d["avg_us_under_counters"] = round(1e3 * r.get("total_ms", 0.0) / n, 1)
        fr, wr = fetch.get((r["kernel"], r["grid"])), write.get((r["kernel"], r["grid"]))
        if fr and wr:
            rd = fr.get("FETCH_SIZE", 0.0) / max(fr["dispatches"], 1) * 1024 * 2
            wb = wr.get("WRITE_SIZE", 0.0) / max(wr["dispatches"], 1) * 1024
            d["hbm_read_bytes"], d["hbm_write_bytes"] = rd, wb
            us = 1e3 * fr.get("total_ms", 0.0) / max(fr["dispatches"], 1)
            if us > 0:
                d["hbm_gbps"] = round((rd + wb) / us / 1e3, 1)
        modes.append(d)
    if modes:
        out["modes"] = modes
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("rows", "modes")}, indent=1))
    for d in out.get("rows", [])[:24]:
        print(d)
    for d in modes:
        print(d)


if __name__ == "__main__":
    main()
