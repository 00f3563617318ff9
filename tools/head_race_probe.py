"""The sampler's output layer (BatchNorm -> relu -> SN 1x1 conv: ops.HeadFn) forward + backward, many times on the same inputs, while a
second process keeps the GPU busy: are dx and the parameter gradients bit-identical every time?  (tools/det_probe_ddp.py found the
first run-to-run difference of a shared-GPU training step in the gradient this layer hands to up_g4.)"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S
from skillful_nowcasting_amd import nn as N

contend = os.environ.get("PROBE_CONTEND", "1") == "1"
proc = None
if contend:
    proc = subprocess.Popen([sys.executable, "-c", "import torch,time\na=torch.randn(4096,4096,device='cuda')\nt=time.time()\n"
                             "while time.time()-t<70:\n    b=a@a\n    b=torch.relu(b)*1e-3\n    torch.cuda.synchronize()"])
    time.sleep(8)
S.set_precision("mixed")
for shape, calls in (((8, 24, 64, 64), 4), ((32, 24, 64, 64), 4), ((96, 48, 128, 128), 6)):
    torch.manual_seed(1)
    bn = N.BatchNorm(shape[1]).cuda().train()
    conv = N.SNConv(shape[1], 4, 1).cuda().train()
    x0 = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(shape[0], 4, shape[2], shape[3], device="cuda").contiguous(memory_format=torch.channels_last)
    sd_bn = {k: v.clone() for k, v in bn.state_dict().items()}
    sd_cv = {k: v.clone() for k, v in conv.state_dict().items()}
    ref, n_bad, worst = None, 0, 0.0
    iters = int(os.environ.get("PROBE_ITERS", "150"))
    for it in range(iters):
        bn.load_state_dict(sd_bn)
        conv.load_state_dict(sd_cv)
        for p in list(bn.parameters()) + list(conv.parameters()):
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = conv(x, bn=bn.prepare(x, calls, None, None), calls=calls, layout=None)
        y.backward(dy)
        torch.cuda.synchronize()
        got = {"y": y.detach().clone(), "dx": x.grad.clone()}
        got.update({"g." + n: p.grad.clone() for n, p in list(bn.named_parameters()) + list(conv.named_parameters()) if p.grad is not None})
        if ref is None:
            ref = got
            continue
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        if bad:
            n_bad += 1
            e = max((ref[k] - got[k]).abs().max().item() / max(ref[k].abs().max().item(), 1e-30) for k in bad)
            worst = max(worst, e)
            if n_bad <= 3:
                d = (ref["dx"] - got["dx"]).abs()
                print(f"  iteration {it}: differ {bad}; dx: {(d > 0).sum().item()} elements, rel {e:.2e}")
    print(f"shape {shape} calls {calls}: {n_bad} of {iters - 1} repetitions differ from the first (worst {worst:.2e})", flush=True)
if proc is not None:
    proc.kill()
