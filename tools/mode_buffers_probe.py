"""Buffers (BatchNorm running statistics, spectral-norm u / v) after n training steps at the paper configuration in two arithmetic
modes from the same state and seeds: which ones deviate, in units of their natural scale."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S

KW = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "bf16x3"]
torch.manual_seed(0)
model = S.DGMR(**KW)
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.to("cuda")
torch.manual_seed(5)
x, y = torch.rand(2, 4, 1, 256, 256).cuda(), torch.rand(2, 18, 1, 256, 256).cuda()
runs = {}
for precision in modes:
    model.load_state_dict(sd0)
    S.ops.bump_weights_epoch()
    model.train()
    model._optimizers = model.configure_optimizers()[0]
    S.ops._NO_PHASES = precision.endswith("-nophase")  # the upsampling convs as written (no phase / pooled / pair-sum decomposition)
    S.set_precision(precision.split("-")[0])
    torch.manual_seed(9)
    losses = []
    for i in range(steps):
        o = model.training_step((x, y), i)
        losses.append([float(o["d_loss"]), float(o["g_loss"]), float(o["grid_loss"])])
    torch.cuda.synchronize()
    S.set_precision("f32")
    runs[precision] = (losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()
                                if k.endswith(("._u", "._v", "running_mean", "running_var")) and not k.startswith(("sampler.", "conditioning_stack.", "latent_stack."))})
    print(precision, losses)
(_, a), (_, b) = runs[modes[0]], runs[modes[1]]
rows = []
for k, ref in a.items():
    if k.endswith("running_mean"):
        scale = a[k[:-len("running_mean")] + "running_var"].sqrt().max().item()
    else:
        scale = ref.abs().max().item()
    rows.append(((b[k] - ref).abs().max().item() / max(scale, 1e-12), k))
rows.sort(reverse=True)
for e, k in rows[:25]:
    print(f"{e:10.3e}  {k}")
