"""Bitwise run-to-run determinism of ops.conv forward / backward at one shape, with the allocator's memory scribbled in between
(GPU box).  usage: python tools/conv_determinism.py [n h w cin cout k groups relu]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import skillful_nowcasting_amd as S  # noqa: E402
from skillful_nowcasting_amd import ops  # noqa: E402


def scribble(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    junk = [torch.randn(1 << 22, device="cuda", generator=g) * 1e3 for _ in range(24)]
    ws = ops._splitk_ws(torch.device("cuda", 0))
    ws.copy_(torch.randn(ws.numel(), device="cuda", generator=g) * 1e3)
    del junk


def main():
    a = [int(v) for v in sys.argv[1:]] or [40, 8, 8, 384, 384, 3, 5, 1]
    n, h, w, cin, cout, k, groups, relu = a
    torch.manual_seed(0)
    x = torch.randn(n, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter((torch.randn(cout, cin, k, k, device="cuda") * (cin * k * k) ** -0.5).contiguous(memory_format=torch.channels_last))
    b = torch.nn.Parameter(torch.randn(cout, device="cuda"))
    kk = cin * k * k
    u = torch.nn.functional.normalize(torch.randn(groups, cout, device="cuda"), dim=1)
    v = torch.nn.functional.normalize(torch.randn(groups, kk, device="cuda"), dim=1)
    inv_sigma = torch.rand(groups, device="cuda") + 0.5
    cot = torch.randn(n, cout, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = []
    for rep in range(6):
        scribble(rep)
        xs = x.clone().requires_grad_(True)
        wt.grad = None
        b.grad = None
        sn = ops.SNCall(inv_sigma, u, v, groups)
        y = ops.conv(xs, wt, b, inv_sigma, None, ops.ConvSpec(pre_relu=bool(relu), sn=sn))
        (y * cot).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), xs.grad.clone(), wt.grad.clone(), b.grad.clone()))
    for i, name in enumerate(("forward", "input grad", "weight grad", "bias grad")):
        d = max((o[i] - outs[0][i]).abs().max().item() for o in outs[1:]) / outs[0][i].abs().max().item()
        print(f"  {name:12s} max run-to-run rel diff {d:.2e}")


if __name__ == "__main__":
    main()
