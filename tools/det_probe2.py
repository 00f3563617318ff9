"""Forward-only determinism probe: the same seeded generator (and discriminator) forwards twice, buffers compared after every forward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
KW = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)


def run(draws):
    S.set_precision(prec)
    torch.manual_seed(7)
    model = S.DGMR(**KW).to("cuda")
    torch.manual_seed(8)
    x = torch.rand(2, 4, 1, 128, 128, device="cuda")
    y = torch.rand(2, 2, 1, 128, 128, device="cuda")
    torch.manual_seed(9)
    snaps = []
    for i in range(4):
        with torch.no_grad():
            out = model.generator.forward_draws(x, draws)
            seq = torch.cat([x, out[:2]], 1)
            sc = model.discriminator(seq)
        torch.cuda.synchronize()
        snap = {"out": out.clone().cpu(), "score": sc.clone().cpu()}
        snap.update({k: v.detach().clone().cpu() for k, v in model.state_dict().items()})
        snaps.append(snap)
    return snaps


for draws in (1, 2, 1):
    a, b, c = run(draws), run(draws), run(draws)
    for name, p_, q_ in (("a-b", a, b), ("b-c", b, c)):
        for i, (sa, sb) in enumerate(zip(p_, q_)):
            bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
            e = (sa["out"].double() - sb["out"].double()).abs().max().item()
            print(f"draws {draws} {name} forward {i}: {len(bad)} of {len(sa)} differ; out max diff {e:.2e}", [k.replace("parametrizations.weight.0.", "") for k in bad[:6]])
