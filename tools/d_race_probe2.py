"""Follow-up to d_race_probe.py: four runs of the temporal discriminator from the same state; for intermediate_dblocks.1 print which
runs agree bitwise on (a) every SNConv's forward output, (b) the gradient at every SNConv output, and where the first differing
elements are."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skillful_nowcasting_amd as S  # noqa: E402
from skillful_nowcasting_amd.common import DBlock  # noqa: E402
from skillful_nowcasting_amd.nn import SNConv  # noqa: E402


def run(td, sd0, seq, cot):
    td.load_state_dict(sd0)
    S.ops.bump_weights_epoch()
    for p in td.parameters():
        p.grad = None
    cap, hooks = {}, []

    def hook(name):
        def h(mod, inp, out):
            cap["fwd " + name] = out.detach().clone()
            out.register_hook(lambda g, nm=name: cap.__setitem__("dout " + nm, g.detach().clone()))
        return h

    for name, m in td.named_modules():
        if isinstance(m, (DBlock, SNConv)):
            hooks.append(m.register_forward_hook(hook(name)))
    out = td(seq)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    return cap


def main():
    torch.manual_seed(0)
    model = S.DGMR(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384).to("cuda").train()
    td = model.discriminator.temporal_discriminator
    sd0 = {k: v.detach().clone() for k, v in td.state_dict().items()}
    torch.manual_seed(31)
    seq = torch.rand(8, 22, 1, 256, 256, device="cuda")
    cot = torch.randn(8, 1, 1, device="cuda")
    runs = [run(td, sd0, seq, cot) for _ in range(4)]
    for k in runs[0]:
        eq = [[bool(torch.equal(runs[i][k], runs[j][k])) for j in range(4)] for i in range(4)]
        if all(all(r) for r in eq):
            continue
        d01 = (runs[0][k] - runs[1][k]).abs()
        nz = int((d01 > 0).sum())
        idx = torch.nonzero(d01 > 0.01 * runs[0][k].abs().max())[:6].tolist()
        print(f"{k:60s} equal-matrix {[''.join('=' if e else 'x' for e in r) for r in eq]}  differing elems(0 vs 1) {nz} of {d01.numel()}  "
              f"max {d01.max().item():.3e} of {runs[0][k].abs().max().item():.3e}; big ones at {idx}")


if __name__ == "__main__":
    main()
