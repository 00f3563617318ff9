"""Bisect a run-to-run difference of the generator forward: forward hooks record every module's output (in execution order) in two
identical runs; the first modules whose outputs are not bit-identical are printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import skillful_nowcasting_amd as S

prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KW = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)


def run():
    S.set_precision(prec)
    torch.manual_seed(7)
    model = S.DGMR(**KW).to("cuda")
    torch.manual_seed(8)
    x = torch.rand(2, 4, 1, 128, 128, device="cuda")
    torch.manual_seed(9)
    rec = []
    names = {m: n for n, m in model.generator.named_modules()}

    def hook(m, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for j, o in enumerate(outs):
            if torch.is_tensor(o):
                rec.append((f"{names[m]}[{j}]", o.detach().float().clone().cpu()))

    hs = [m.register_forward_hook(hook) for m in model.generator.modules()]
    with torch.no_grad():
        out = model.generator.forward_draws(x, draws)
    torch.cuda.synchronize()
    for h in hs:
        h.remove()
    rec.append(("OUT", out.clone().cpu()))
    return rec


contender = None
if os.environ.get("PROBE_CONTEND") == "1":  # a second process keeps the GPU busy (what two DDP ranks on one GPU do to each other)
    import subprocess

    contender = subprocess.Popen([sys.executable, "-c", "import torch,time\na=torch.randn(4096,4096,device='cuda')\nt=time.time()\n"
                                  "while time.time()-t<float(%r):\n    b=a@a\n    b=torch.relu(b)*1e-3\n    torch.cuda.synchronize()" % os.environ.get("PROBE_CONTEND_S", "60")])
    import time

    time.sleep(8)
N = int(os.environ.get("PROBE_RUNS", "3"))
runs = [run() for _ in range(N)]
if contender is not None:
    contender.kill()
for (na, a), (nb, b) in [((f"run0", runs[0]), (f"run{i}", runs[i])) for i in range(1, N)]:
    print(f"--- {na} vs {nb}: {len(a)} recorded outputs")
    shown = 0
    for (ka, ta), (kb, tb) in zip(a, b):
        assert ka == kb
        if not torch.equal(ta, tb):
            e = (ta.double() - tb.double()).abs().max().item()
            print(f"  DIFF {ka:60s} shape {tuple(ta.shape)} max diff {e:.2e} of max {tb.abs().max().item():.2e}")
            shown += 1
            if shown >= 12:
                break
    if not shown:
        print("  all identical")
