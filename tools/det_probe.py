"""Where do two identical runs part?  python tools/det_probe.py [precision] [steps] [paper]   (GPU only; environment switches such as
DGMR_SN_PREFETCH=0 / DGMR_WGRAD_STREAM=0 / DGMR_BRANCH_STREAM=0 select the suspects).  Runs the same seeded training twice, snapshots
every parameter, buffer and GRADIENT after every step and prints, per step, the tensors that are not bit-identical."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
paper = len(sys.argv) > 3 and sys.argv[3] == "paper"
KW = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6) if paper else \
    dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
B = 1 if paper else 2


def run():
    S.set_precision(prec)
    torch.manual_seed(7)
    model = S.DGMR(**KW).to("cuda")
    torch.manual_seed(8)
    hw = KW["output_shape"]
    x = torch.rand(B, 4, 1, hw, hw, device="cuda")
    y = torch.rand(B, KW["forecast_steps"], 1, hw, hw, device="cuda")
    torch.manual_seed(9)
    snaps = []
    for i in range(steps):
        out = model.training_step((x, y), i)
        torch.cuda.synchronize()
        snap = {"loss." + k: v.detach().clone().cpu() for k, v in out.items()}
        snap.update({"state." + k: v.detach().clone().cpu() for k, v in model.state_dict().items()})
        snap.update({"grad." + k: p.grad.detach().clone().cpu() for k, p in model.named_parameters() if p.grad is not None})
        snaps.append(snap)
    return snaps


print("deterministic:", S.deterministic(), "precision:", prec, {k: v for k, v in os.environ.items() if k.startswith("DGMR_")})
a, b = run(), run()
for i, (sa, sb) in enumerate(zip(a, b)):
    bad = [(k, (sa[k].double() - sb[k].double()).abs().max().item() / max(sb[k].double().abs().max().item(), 1e-300)) for k in sa
           if k in sb and not torch.equal(sa[k], sb[k])]
    kinds = {}
    for k, _ in bad:
        kinds[k.split(".")[0]] = kinds.get(k.split(".")[0], 0) + 1
    print(f"step {i}: {len(bad)} of {len(sa)} tensors differ {kinds}")
    for k, e in bad[:14]:
        print(f"    {k:90s} rel {e:.2e}  shape {tuple(sa[k].shape)}")
