"""Where the HOST spends a training step (run on the GPU box):  python tools/host_phases.py [steps]

Wraps the phases of DGMR._training_step with host timers (no device synchronisation) and prints, per step, the host milliseconds inside
each phase next to the step's device time.  A phase whose host time grows from the first (free-running) step to the steady state is where
the host waits - for queue space, or for a blocking call.  Used in round 6 to find the blocking `.to(device)` uploads of the latent draws
and frame indices (ops.upload)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
import skillful_nowcasting_amd as S  # noqa: E402
from skillful_nowcasting_amd import ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
S.set_precision("mixed")
torch.manual_seed(0)
model = S.DGMR(forecast_steps=18, output_shape=256).to(dev)
B = 16
images = torch.rand(B, 4, 1, 256, 256, device=dev)
future = torch.rand(B, 18, 1, 256, 256, device=dev)
acc = {}
order = []
calls = []  # (label, start ms since the step began, host ms) of every wrapped call of the current step
step_t0 = [0.0]


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            if label not in order:
                order.append(label)
            acc[label] = acc.get(label, 0.0) + 1e3 * (time.perf_counter() - t0)
            calls.append((label, 1e3 * (t0 - step_t0[0]), 1e3 * (time.perf_counter() - t0)))

    setattr(obj, name, timed)


wrap(model.latent_stack, "draw", "latent draw (upload)")
wrap(model.generator, "forward_draws", "generator.forward_draws")
wrap(model, "_disc_losses", "_disc_losses (D forward)")
wrap(model, "_gen_losses", "_gen_losses (D forward on 6 draws)")
wrap(model, "manual_backward", "manual_backward")
wrap(ops, "join_side_streams", "join_side_streams")
from skillful_nowcasting_amd import nn as snn  # noqa: E402

wrap(snn.SNPlan, "run", "  SNPlan.run")
wrap(snn.SNScope, "_prefetch", "  SNScope._prefetch")
wrap(model.generator.conditioning_stack, "forward", "  conditioning stack (the first launches of a forward: where the host waits for queue space)")
g_opt, d_opt = model.optimizers()
wrap(g_opt, "step", "g_opt.step")
wrap(d_opt, "step", "d_opt.step")
wrap(g_opt, "zero_grad", "zero_grad")
wrap(d_opt, "zero_grad", "zero_grad")
model.optimizers = lambda: (g_opt, d_opt)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
rows = []
torch.cuda.synchronize()
evs[0].record()
for i in range(steps):
    acc.clear()
    calls.clear()
    t0 = step_t0[0] = time.perf_counter()
    model.training_step((images, future), i)
    tot = 1e3 * (time.perf_counter() - t0)
    evs[i + 1].record()
    rows.append((tot, dict(acc), list(calls)))
torch.cuda.synchronize()
print("step | host ms total | device ms | " + " | ".join(order))
for i, (tot, a, _) in enumerate(rows):
    print(f"{i:4d} | {tot:8.1f} | {evs[i].elapsed_time(evs[i + 1]):8.1f} | " + " | ".join(f"{a.get(k, 0.0):8.1f}" for k in order))
print("\nthe last step, call by call (start ms | host ms | phase):")
for label, ts, dur in rows[-1][2]:
    print(f"  {ts:8.1f} | {dur:8.1f} | {label}")
