import sys, os, time
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
print("priority range", torch.cuda.Stream.priority_range())
def run(stream, n=4):
    ts = []
    with torch.cuda.stream(stream):
        for i in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.training_step((x, y), i)
            torch.cuda.synchronize()
            ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    return ts
print("default stream", run(torch.cuda.default_stream()))
hp = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
print("high-priority main", run(hp))
print("default stream", run(torch.cuda.default_stream()))
print("high-priority main", run(hp))
