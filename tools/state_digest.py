"""sha256 of every parameter, buffer and Adam moment after a few seeded training steps (run on the GPU box):

    python tools/state_digest.py [steps=2] [batch=2]          DGMR_LIB=<other build> for the other side of an A/B

Two builds of the library that claim bit-identical results (a kernel rewritten for speed with the same operations in the same order) must
print the same digest.  Paper configuration at a small batch: every kernel class of the step runs, the six draws and the call groups included."""
import hashlib
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

if not __import__("os").environ.get("DGMR_LIB"):
    g.build()
import skillful_nowcasting_amd as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
S.set_precision("mixed")
torch.manual_seed(0)
model = S.DGMR(forecast_steps=18, output_shape=256).to(dev)
torch.manual_seed(1)
x = torch.rand(B, 4, 1, 256, 256, device=dev)
y = torch.rand(B, 18, 1, 256, 256, device=dev)
torch.manual_seed(2)
for i in range(steps):
    out = model.training_step((x, y), i)
torch.cuda.synchronize()
h = hashlib.sha256()
n = 0
for k, v in sorted(model.state_dict().items()):
    h.update(k.encode())
    h.update(v.detach().cpu().contiguous().numpy().tobytes())
    n += 1
for opt in model.optimizers():
    for st in opt.state.values():
        for kk in ("exp_avg", "exp_avg_sq"):
            h.update(st[kk].detach().cpu().contiguous().numpy().tobytes())
print(f"digest {h.hexdigest()}  ({n} state tensors, {steps} steps, batch {B}; losses {[float(v) for v in out.values()]})")
