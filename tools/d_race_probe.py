"""Is the temporal discriminator's backward deterministic?  (GPU box, exact f32)  Runs forward + backward several times from the same
state and inputs and reports the largest run-to-run difference of every block-output gradient and parameter gradient; then repeats
with split-K disabled (dgmr_conv_tune ksplit = 1).  Float atomics (bias gradients) give ~1e-6; anything near 1e-3 is a race or a
read of uninitialised memory."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skillful_nowcasting_amd as S  # noqa: E402
from skillful_nowcasting_amd._lib import call  # noqa: E402
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
    for n, p in td.named_parameters():
        if p.grad is not None:
            cap["grad " + n] = p.grad.detach().clone()
    return cap


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "temporal"
    torch.manual_seed(0)
    model = S.DGMR(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384).to("cuda").train()
    td = getattr(model.discriminator, which + "_discriminator")
    sd0 = {k: v.detach().clone() for k, v in td.state_dict().items()}
    torch.manual_seed(31)
    seq = torch.rand(8, 22, 1, 256, 256, device="cuda")
    cot = torch.randn(8, 1, 1, device="cuda")
    for label, tune in (("default dispatch", (-1, -1, -1, -1)),):
        call("dgmr_conv_tune", *tune)
        torch.manual_seed(3)
        base = run(td, sd0, seq, cot)
        worst = {}
        for _ in range(3):
            torch.manual_seed(3)
            other = run(td, sd0, seq, cot)
            for k, v in base.items():
                d = (other[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30)
                worst[k] = max(worst.get(k, 0.0), d)
        print(f"== {which}: {label}: run-to-run differences (max over 3 repeats) ==")
        for k, d in worst.items():
            if d > 3e-5:
                print(f"  {k:70s} {d:.2e}")
    call("dgmr_conv_tune", -1, -1, -1, -1)


if __name__ == "__main__":
    main()
