"""bench_detail.json (the full record bench.py writes beside its one-line summary) -> a markdown table of the per-kernel rows.
    python tools/roofline_report.py profiles/r05_final_B16_bench_detail.json > profiles/r05_final_roofline_table.md"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"# {d['metric']} - {d['config']['workload']}\n")
print(f"{d['ms_per_step']:.1f} ms/step = {d['value']:.1f} {d['unit']} ({d['dtype']}), {d['steps']} steps after {d['warmup']} warm-ups, per-GPU batch "
      f"{d['config']['per_gpu_batch']}; whole step {r['whole_step']['tflops']:.0f} TFLOP/s algorithmic = {r['whole_step']['frac']:.3f} of the {r['peak']:.0f} TF bf16 peak.\n")
print(f"Dominant class `{r['kernel']}`: {r['launches_per_step']} launches, {r['achieved']:.0f} TFLOP/s = **{r['frac']:.3f}** of peak "
      f"(executed MFMA rate {r['mfma_executed_frac']:.3f}); time-weighted matrix-pipe busy over all 3x3 window launches (PMC): "
      f"{r.get('mfma_util_weighted')}.\n")
print("Rows per INSTANTIATED kernel x mode x launch size, one extra step with HIP events around every launch (weight gradients in line):\n")
print("| kernel | launches | total ms | avg us | TFLOP/s (algorithmic) | executed MFMA TFLOP/s | frac of peak |")
print("|---|---:|---:|---:|---:|---:|---:|")
for row in r["per_kernel_detail"][:40]:
    print(f"| `{row['kernel']}` | {row['launches']} | {row['total_ms']:.1f} | {row['avg_us']:.0f} | {row['tflops']:.0f} | {row['mfma_executed_tflops']:.0f} | {row['frac']:.3f} |")
tot = sum(x["total_ms"] for x in r["per_kernel_detail"])
print(f"\nAll {len(r['per_kernel_detail'])} rows: {tot:.1f} ms of conv kernels per step (in line).")
for mode, leg in (d.get("also") or {}).items():
    extra = ""
    if leg.get("roofline"):
        extra = f"; dominant class `{leg['roofline']['kernel']}` {leg['roofline']['achieved']:.1f} TF = {leg['roofline']['frac']:.3f} of {leg['roofline']['peak']:.1f} TF"
    print(f"\n* `{mode}`: {leg['ms_per_step']:.1f} ms/step = {leg['radar_frames_per_s']:.1f} frames/s ({leg['steps']} steps){extra}")
cb = d.get("cpu_baseline")
if cb:
    print(f"\nCPU baseline ({cb['kind']}): {cb['value']:.3f} {cb['unit']} on {cb['cores']} of {cb['host_cpus']} host threads - {cb['sample']}")
