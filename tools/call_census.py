"""Host-side census of C-ABI calls in one training step (GPU only): how many launches each call site issues.

    python tools/call_census.py [--workload=paper] [--batch=16] [--top=40]

Wraps skillful_nowcasting_amd.ops.call for one steady-state step and prints (entry point, caller file:line) -> count, plus
the host time of the step with the GPU idle-free (no sync inside).
"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S
from skillful_nowcasting_amd import ops

WORKLOADS = {
    "paper": (dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6), 256),
    "smoke": (dict(forecast_steps=4, output_shape=128, latent_channels=384, context_channels=192, generation_steps=3), 128),
}


def main():
    wl, batch, top = "paper", 16, 40
    for a in sys.argv[1:]:
        if a.startswith("--workload="):
            wl = a.split("=")[1]
        if a.startswith("--batch="):
            batch = int(a.split("=")[1])
        if a.startswith("--top="):
            top = int(a.split("=")[1])
    kw, hw = WORKLOADS[wl]
    ops.set_precision("bf16x3")
    torch.manual_seed(0)
    model = S.DGMR(**kw).to("cuda")
    t = kw["forecast_steps"]
    b = (torch.rand(batch, 4, 1, hw, hw, device="cuda"), torch.rand(batch, t, 1, hw, hw, device="cuda"))
    model.training_step(b, 0)
    torch.cuda.synchronize()
    counts = collections.Counter()
    orig = ops.call

    def counting(name, *args):
        f = sys._getframe(1)
        if f.f_code.co_name in ("_copy", "_launch_conv"):
            f = f.f_back
        counts[(name, f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno} {f.f_code.co_name}")] += 1
        return orig(name, *args)

    ops.call = counting
    t0 = time.perf_counter()
    model.training_step(b, 1)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ops.call = orig
    total = sum(counts.values())
    print(f"{total} C-ABI calls from ops.py in one step; host {1e3 * host:.0f} ms, wall {1e3 * wall:.0f} ms")
    by_name = collections.Counter()
    for (n, _), c in counts.items():
        by_name[n] += c
    print("by entry point:", ", ".join(f"{n}={c}" for n, c in by_name.most_common(12)))
    for (n, site), c in counts.most_common(top):
        print(f"  {c:6d}  {n:28s} {site}")


if __name__ == "__main__":
    main()
