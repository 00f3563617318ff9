import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import skillful_nowcasting_amd as S
KW = dict(forecast_steps=2, output_shape=128, latent_channels=384, context_channels=192, generation_steps=2)
print("threads", torch.get_num_threads())
sds = []
for r in range(4):
    torch.manual_seed(7)
    m = S.DGMR(**KW)
    sds.append({k: v.clone() for k, v in m.state_dict().items()})
for r in range(1, 4):
    bad = [k for k in sds[0] if not torch.equal(sds[0][k], sds[r][k])]
    print("construction", r, len(bad), bad[:6])
