"""Derived numbers for the dominant conv kernel from the PMC passes of tools/pmc_conv.sh.

    python tools/pmc_derive.py <concatenated pmc_pass*.csv> <kernel substring> <grid> <launch_us> <out.json> key=value ...

key=value pairs: label (bench.py's variant label), shape (free text), precision, alg_flops, alg_bytes.
Formulas (MI355X_MICROARCH.md, HBM / rocprofv3 section):
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  HBM read    = FETCH_SIZE [KB] x 1024 x 2   (gfx950 counts a 128-byte request as 64 bytes for wide coalesced reads)
  HBM write   = WRITE_SIZE [KB] x 1024
"""
import csv
import json
import sys


def main():
    src, ksub, grid, launch_us, dst = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
    kv = dict(a.split("=", 1) for a in sys.argv[6:])
    c = {}
    name = ""
    with open(src) as f:
        for r in csv.reader(f):
            if len(r) != 5 or r[0] == "kernel":
                continue
            if ksub in r[0] and r[2] == grid:
                c[r[1]] = float(r[4])
                name = r[0]
    need = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]
    missing = [k for k in need if k not in c]
    if missing:
        raise SystemExit(f"missing counters {missing} for kernel '{ksub}' grid {grid}; have {sorted(c)}")
    alg_flops, alg_bytes = float(kv["alg_flops"]), float(kv["alg_bytes"])
    mult = {"f32": 1, "bf16x3": 3, "bf16": 1}[kv.get("precision", "bf16x3")]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
    out = {
        "kernel": f"{kv.get('label', '')} ({name.split('::')[-1].split('(')[0]})",
        "precision": kv.get("precision", "bf16x3"),
        "shape": kv.get("shape", ""),
        "launch_us": launch_us,
        "algorithmic_tflops": alg_flops / launch_us / 1e6,
        # matrix-pipe work from the instruction counter when it was collected (32x32x16 bf16 MFMA = 32768 flop per wave-instruction):
        # exact also for launches that execute fewer MACs than the algorithmic count (phase / pooled upsampling convs)
        "mfma_executed_tflops": (c["SQ_INSTS_MFMA"] * 32768.0 if c.get("SQ_INSTS_MFMA", 0) > 0 and kv.get("precision", "bf16x3") != "f32"
                                 else mult * alg_flops) / launch_us / 1e6,
        "gpu_cycles_per_xcd": cyc,
        "effective_clock_ghz": cyc / launch_us / 1e3,
        "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
        "mfma_util_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs): share of the SIMD cycles of this launch in "
                          "which the matrix pipe was busy (counter passes run the kernel ~5-10 % slower than the un-instrumented launch_us)",
        "mfma_executed_frac_of_spec_peak": (c["SQ_INSTS_MFMA"] * 32768.0 if c.get("SQ_INSTS_MFMA", 0) > 0 and kv.get("precision", "bf16x3") != "f32"
                                            else mult * alg_flops) / launch_us / 1e6 / 2500.0,
        "fetch_size_kb": c["FETCH_SIZE"],
        "write_size_kb": c["WRITE_SIZE"],
        "hbm_read_bytes_corrected": rd,
        "hbm_write_bytes": wr,
        "traffic_bytes": rd + wr,
        "traffic_note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads); "
                        "WRITE_SIZE x 1024",
        "algorithmic_bytes": alg_bytes,
        "traffic_over_algorithmic": (rd + wr) / alg_bytes,
        "hbm_gbps": (rd + wr) / launch_us / 1e3,
    }
    if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"] > 0:
        out["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"]
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
        out["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        out["lds_active_frac_of_cu_cycles"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
    if "SQ_WAVE_CYCLES" in c and all(k in c for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        tot = c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]
        out["wave_cycles_split"] = {"wait_any": c["SQ_WAIT_ANY"] / tot, "wait_inst_any": c["SQ_WAIT_INST_ANY"] / tot,
                                    "active_inst_any": c["SQ_ACTIVE_INST_ANY"] / tot}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
