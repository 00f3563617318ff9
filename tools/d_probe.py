"""Where does a discriminator backward lose accuracy?  (GPU box)

Runs the paper-size discriminator forward + backward on N sequences twice - exact `f32` arithmetic and another mode (default bf16x3)
- from the same weights, inputs and frame indices, and prints per spectral-norm conv, in execution order: the relative difference
of its output (forward), of the gradient arriving at its output (backward) and of its weight gradient.  The first row where a
column jumps is the op to look at.  No oracle involved (mode vs mode of the same package).
    python tools/d_probe.py [--mode bf16x3] [--n 8] [--which spatial|temporal|both]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skillful_nowcasting_amd as S  # noqa: E402
from skillful_nowcasting_amd.nn import SNConv, SNLinear1  # noqa: E402


def run(disc, x, cot, mode, seed):
    S.set_precision(mode)
    for p in disc.parameters():
        p.grad = None
    rec = {}
    hooks = []

    def fwd_hook(name):
        def h(mod, inp, out):
            rec[name + ".out"] = out.detach().clone()
            if out.requires_grad:
                out.register_hook(lambda g, n=name: rec.__setitem__(n + ".dout", g.detach().clone()))
        return h

    for name, m in disc.named_modules():
        if isinstance(m, (SNConv, SNLinear1)):
            hooks.append(m.register_forward_hook(fwd_hook(name)))
    xs = x.clone().requires_grad_(True)
    torch.manual_seed(seed)
    out = disc(xs)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    rec["scores"] = out.detach().clone()
    rec["dx"] = xs.grad.detach().clone()
    for name, p in disc.named_parameters():
        if p.grad is not None:
            rec["grad." + name] = p.grad.detach().clone()
    S.set_precision("f32")
    return rec


def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16x3")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--which", default="both")
    args = ap.parse_args()
    torch.manual_seed(0)
    model = S.DGMR(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384).to("cuda").train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(31)
    x = torch.rand(args.n, 22, 1, 256, 256, device="cuda")
    for which in (["spatial", "temporal"] if args.which == "both" else [args.which]):
        disc = getattr(model.discriminator, which + "_discriminator")
        cot = torch.randn(args.n, 1, 1, device="cuda")
        res = {}
        for mode in ("f32", args.mode):
            model.load_state_dict(sd0)
            S.ops.bump_weights_epoch()
            res[mode] = run(disc, x, cot, mode, 3)
        a, b = res[args.mode], res["f32"]
        print(f"\n== {which} discriminator, {args.mode} vs f32, N={args.n} ==")
        print(f"  {'scores':70s} {rel(a['scores'], b['scores']):.2e}")
        print(f"  {'d / d frames':70s} {rel(a['dx'], b['dx']):.2e}")
        print(f"  {'module':58s} {'shape of out':>24s} {'fwd':>9s} {'d(out)':>9s} {'d(weight)':>9s}")
        for name, m in disc.named_modules():
            if not isinstance(m, (SNConv, SNLinear1)):
                continue
            f = rel(a[name + ".out"], b[name + ".out"])
            g = rel(a[name + ".dout"], b[name + ".dout"]) if name + ".dout" in a else float("nan")
            wk = "grad." + name + ".parametrizations.weight.original"
            w = rel(a[wk], b[wk]) if wk in a else float("nan")
            print(f"  {name:58s} {str(tuple(b[name + '.out'].shape)):>24s} {f:9.2e} {g:9.2e} {w:9.2e}")


if __name__ == "__main__":
    main()
