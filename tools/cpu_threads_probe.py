"""How the CPU oracle scales with torch threads on this host (picks the thread count bench.py's cpu_baseline leg should use)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skillful_nowcasting_amd as S
from oracle import dgmr_oracle as O

kw = dict(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384, generation_steps=6)
torch.manual_seed(0)
model = S.DGMR(**kw)
sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith(("generator.", "discriminator."))}
del model
x = torch.rand(1, 4, 1, 256, 256)
z = O.draw_latent((8, 8, 8))
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.generator(sd, "generator.", x, z, 18, True)
        t0 = time.perf_counter()
        O.generator(sd, "generator.", x, z, 18, True)
        dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: G forward {dt:.2f} s", flush=True)
