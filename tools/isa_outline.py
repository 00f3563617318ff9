"""Outline of a kernel's ISA between barriers: python tools/isa_outline.py file.s <kernel-substring> [--full]
Per barrier-delimited segment: counts of MFMA / VALU / LDS / global loads / stores / LDS-DMA and every s_waitcnt with a vmcnt field
(in order, with the number of VMEM instructions issued since the previous one) - how the compiler ordered loads and waits."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ": ;" in l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    seg = dict(mfma=0, valu=0, ds_r=0, ds_w=0, gld=0, gst=0, dma=0, salu=0)
    waits = []
    since = 0
    nseg = 0
    for l in lines[start:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")):
            if t.startswith(".LBB"):
                print(f"      {t}")
            continue
        op = t.split()[0]
        if op == "s_barrier":
            print(f"seg {nseg:3d}: " + " ".join(f"{k}={v}" for k, v in seg.items() if v) + ("  waits: " + " ".join(waits) if waits else ""))
            seg = dict.fromkeys(seg, 0)
            waits = []
            nseg += 1
            continue
        if op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                waits.append(f"vm{m.group(1)}(+{since})")
                since = 0
            continue
        if op.startswith("v_mfma"):
            seg["mfma"] += 1
        elif op.startswith("global_load_lds") or ("lds" in t and op.startswith("buffer_load")):
            seg["dma"] += 1
            since += 1
        elif op.startswith(("global_load", "buffer_load", "flat_load")):
            seg["gld"] += 1
            since += 1
        elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
            seg["gst"] += 1
            since += 1
        elif op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
            seg["ds_r"] += 1
        elif op.startswith("ds_write"):
            seg["ds_w"] += 1
        elif op.startswith("v_"):
            seg["valu"] += 1
        elif op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_nop", "s_endpgm")):
            seg["salu"] += 1
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")) and "--full" in sys.argv:
            print(f"      {t}")
    print(f"tail   : " + " ".join(f"{k}={v}" for k, v in seg.items() if v) + ("  waits: " + " ".join(waits) if waits else ""))


main()
