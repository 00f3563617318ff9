"""The bench contract's ONE JSON line must stay small enough for the driver to parse (round 4's 52 KB line left BENCH_r04.parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _synthetic(n_rows=400, n_classes=40):
    row = lambda i: dict(kernel=f"conv_fwd_dgrad_win3x3<128px,{i}> plain big [bf16x3] " + "x" * 40, launches=i + 1, total_ms=1000.0 / (i + 1),
                         avg_us=123.456789, flops_per_launch=1.23456789e11, tflops=321.123456, mfma_executed_tflops=963.3, peak_tflops=2500.0,
                         frac=0.12845)
    roof = {"bound": "mfma", "achieved": 366.04, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.1464, "traffic": 6.26e9, "kernel": "conv_fwd_dgrad_win3x3<128px,96>",
            "traffic_over_algorithmic": 1.15, "mfma_util": 0.49, "mfma_util_weighted": 0.41, "valu_per_mfma": 4.98, "mfma_executed_frac": 0.35,
            "launches_per_step": 557, "avg_launch_us": 487.2, "flops_per_launch": 1.783e11, "note": "n" * 500,
            "whole_step": {"tflops": 213.8, "frac": 0.0855, "note": "w" * 300}, "all_conv_kernels": {"tflops": 319.8, "ms_per_step": 780.9, "frac_of_step": 0.9},
            "per_kernel": [row(i) for i in range(n_classes)], "per_kernel_detail": [row(i) for i in range(n_rows)],
            "pmc_launch": {"shape": "s" * 900}, "mfma_util_classes": {"rows": [row(i) for i in range(20)]}}
    return {
        "metric": "radar frames/sec (G+D step) 4->18 @256^2", "value": 409.123456, "unit": "radar frames/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 860.512345, "ms_per_step_median": 859.8, "step_ms": [860.5] * 20, "step_host_ms__reserved_gb__device_allocs__retries": [(1.0, 2.0, 3, 0)] * 20,
        "hbm": {"max_allocated_gb": 105.1, "reserved_gb": 110.9, "alloc_retries": 0}, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": bench.DTYPE_SHORT["mixed"], "dtype_long": bench.DTYPE_TEXT["mixed"], "data": "synthetic torch.rand frames, random-init weights",
        "config": {"workload": "DGMR.training_step paper: forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6, H=W=256",
                   "per_gpu_batch": 16, "global_batch": 16, "frames_per_sample": 22, "parallelism": "dp1", "semantics": "strict: " + "s" * 120},
        "roofline": roof,
        "also": {"f32": {"ms_per_step": 3553.0, "radar_frames_per_s": 99.07, "steps": 10, "step_ms": [3553.0] * 10, "roofline": dict(roof, peak=157.3, achieved=78.9, frac=0.5016)},
                 "bf16x6": {"ms_per_step": 1431.0, "radar_frames_per_s": 246.0, "steps": 5}, "bf16": {"ms_per_step": 579.9, "radar_frames_per_s": 607.0, "steps": 5}},
        "cpu_baseline": {"value": 0.79, "unit": "radar frames/s", "cores": 16, "kind": "reference", "host_cpus": 256,
                         "seconds_per_step": {"as_written_anomaly_on": 36.9, "anomaly_off": 27.9}, "sample": "t" * 400,
                         "reference_measured": {"what": "r" * 300}},
        "process_group": {"backend": "nccl", "world_size_reported": 8, "rccl_version": "2.26.6", "ranks_seen": [{"rank": i, "device": "d" * 60} for i in range(8)],
                          "grad_sync": {"overlapped_buckets": 5, "late_buckets": 2, "deviations": 0}},
    }


def test_line_is_bounded_and_round_trips():
    full = _synthetic()
    assert len(json.dumps(full)) > 50_000  # the record that broke round 4's parse
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line) < 6000, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"].startswith("DGMR.training_step paper")
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "mfma_util_weighted"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert len(r["top_rows"]) <= 8 and "per_kernel_detail" not in r
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 16 and cb["value"] > 0 and "sample" in cb
    assert d["also"]["f32"]["roofline"]["peak"] == 157.3
    assert d["detail"] == "gpurun_out/bench_detail.json"


def test_line_drops_optional_blocks_rather_than_overflow():
    full = _synthetic()
    full["config"]["semantics"] = "s" * 2500
    full["cpu_baseline"]["sample"] = "t" * 3000
    line = bench.compact_line(full, "x.json")
    assert len(line) <= bench.LINE_LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"] and d["value"]


def test_line_bound_holds_for_arbitrary_records():
    """Property test (hypothesis): whatever the per-kernel tables, notes and process-group records look like, the line stays within the
    limit, stays one line of valid JSON and keeps the headline, the roofline fraction and the CPU baseline."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    text = st.text(alphabet=st.characters(min_codepoint=32, max_codepoint=126), max_size=4000)

    @settings(max_examples=60, deadline=None)
    @given(n_rows=st.integers(0, 600), n_classes=st.integers(1, 80), note=text, sample=text, workload=st.text(alphabet="abcdefgh =,", max_size=400),
           ranks=st.integers(0, 64), ms=st.floats(1e-3, 1e6, allow_nan=False))
    def check(n_rows, n_classes, note, sample, workload, ranks, ms):
        full = _synthetic(n_rows, n_classes)
        full["ms_per_step"] = ms
        full["roofline"]["note"] = note
        full["cpu_baseline"]["sample"] = sample
        full["config"]["workload"] = "DGMR.training_step " + workload
        full["process_group"]["ranks_seen"] = [{"rank": i, "device": "d" * 80} for i in range(ranks)]
        line = bench.compact_line(full, "gpurun_out/bench_detail.json")
        assert "\n" not in line and len(line) <= bench.LINE_LIMIT
        d = json.loads(line)
        assert d["metric"] and d["value"] and d["ms_per_step"] > 0 and d["n_gpus"] == 1
        assert d["roofline"]["frac"] and d["roofline"]["bound"] == "mfma"
        assert d["cpu_baseline"]["value"] and d["cpu_baseline"]["kind"] == "reference"

    check()
