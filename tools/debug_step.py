"""GPU debug: per-key gradient errors of the HIP training step vs the reference golden."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_training_step import _golden, _snapshot_grads
import skillful_nowcasting_amd as S

rec, keys, kw = _golden()
torch.manual_seed(42)
model = S.DGMR(**kw).to("cuda")
bw = []
orig = model.manual_backward
model.manual_backward = lambda loss: (bw.append(loss.detach()), orig(loss))
grads = {}
named = {("generator." + k if not k.startswith("discriminator.") else k): p for k, p in model.named_parameters()}
g_opt, d_opt = model.optimizers()
_snapshot_grads(g_opt, named, "generator.", grads, False)
_snapshot_grads(d_opt, named, "discriminator.", grads, True)
torch.manual_seed(44)
out = model.training_step((rec["images"].cuda(), rec["future"].cuda()), 0)
torch.cuda.synchronize()
print("backward losses", [float(x) for x in bw], rec["backward_losses"].tolist())
print("losses", {k: float(v) for k, v in out.items()}, rec["losses"].tolist())
for k, ref in rec.items():
    if not k.startswith("grad."):
        continue
    if k[5:] not in grads:
        print("MISSING", k); continue
    got = grads[k[5:]].detach().cpu().float().reshape(ref.shape)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"{k[5:]:90s} rel err {err / (scale + 1e-30):.3e} scale {scale:.3e}")
