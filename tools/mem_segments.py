import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as ge
ge.build()
import skillful_nowcasting_amd as S
S.set_precision("bf16x3")
torch.manual_seed(0)
model = S.DGMR(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6).to("cuda")
model.train()
x = torch.rand(16, 4, 1, 256, 256, device="cuda"); y = torch.rand(16, 18, 1, 256, 256, device="cuda")
for i in range(3):
    model.training_step((x, y), i)
    torch.cuda.synchronize()
    snap = torch.cuda.memory_snapshot()
    segs = sorted(snap, key=lambda s: -s["total_size"])[:6]
    print("step", i, "reserved %.1f GB" % (torch.cuda.memory_reserved() / 2**30), [round(s["total_size"] / 2**30, 2) for s in segs])
# the biggest segment: what lives in it now
big = sorted(torch.cuda.memory_snapshot(), key=lambda s: -s["total_size"])[0]
print([(round(b["size"] / 2**30, 2), b["state"]) for b in big["blocks"]][:12])
