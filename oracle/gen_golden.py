"""Generate tests/golden/*.safetensors by running the UNMODIFIED reference on CPU (build container only).

Usage:  python oracle/gen_golden.py            (needs /root/reference; writes tests/golden/)

The reference ships no numeric golden vectors for the hot path, so these fixtures —
outputs of the reference's own modules on seeded inputs — are what pins ``oracle/dgmr_oracle.py``.
Each file holds: ``sd0.<key>`` the module state_dict BEFORE the call, ``buf1.<key>`` the buffers
AFTER the call (u/v, BN running stats), ``in.*`` inputs, ``out.*`` outputs, ``grad.*`` gradients of
``sum(out * cot)`` w.r.t. inputs / selected parameters, ``cot`` the cotangent.
Test infrastructure: never imported by the product.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import _stubs  # noqa: E402

_stubs.install()
torch.autograd.set_detect_anomaly(False)

from dgmr.common import (  # noqa: E402
    ContextConditioningStack,
    DBlock,
    GBlock,
    LatentConditioningStack,
    LBlock,
    UpsampleGBlock,
)
from dgmr.discriminators import SpatialDiscriminator, TemporalDiscriminator  # noqa: E402
from dgmr.generators import Sampler  # noqa: E402
from dgmr.layers import AttentionLayer, ConvGRU  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def _is_buffer_key(k):
    return k.endswith(("._u", "._v", "running_mean", "running_var", "num_batches_tracked"))


def record(name, module, inputs, call, train=True, grad_params=True, meta=None):
    """Run `call(module, *inputs)` once; store state before/after, outputs and gradients."""
    module.train(train)
    rec = {}
    for k, v in module.state_dict().items():
        rec["sd0." + k] = v.detach().clone()
    ins = []
    for i, x in enumerate(inputs):
        if isinstance(x, torch.Tensor) and x.is_floating_point():
            x = x.clone().requires_grad_(True)
        ins.append(x)
    out = call(module, *ins)
    outs = out if isinstance(out, (tuple, list)) else [out]
    torch.manual_seed(1234)
    cots = [torch.randn_like(o) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    loss.backward()
    for i, x in enumerate(ins):
        if isinstance(x, torch.Tensor):
            rec[f"in.{i}"] = x.detach().clone()
            if x.is_floating_point() and x.grad is not None:
                rec[f"grad.in.{i}"] = x.grad.clone()
    for i, (o, c) in enumerate(zip(outs, cots)):
        rec[f"out.{i}"] = o.detach().clone()
        rec[f"cot.{i}"] = c
    if grad_params:
        for k, p in module.named_parameters():
            if p.grad is not None:
                rec["grad.p." + k] = p.grad.clone()
    for k, v in module.state_dict().items():
        if _is_buffer_key(k):
            rec["buf1." + k] = v.detach().clone()
    rec = {k: v.contiguous() for k, v in rec.items()}
    md = {"name": name, "train": str(train)}
    if meta:
        md.update({k: json.dumps(v) for k, v in meta.items()})
    save_file(rec, os.path.join(OUT, name + ".safetensors"), metadata=md)
    nbytes = sum(v.numel() * v.element_size() for v in rec.values())
    print(f"{name:28s} {len(rec):4d} tensors {nbytes / 1e6:7.2f} MB")


def main():
    # ---- DBlock 2-D / 3-D (dgmr/common.py:158-238) ----
    torch.manual_seed(1)
    record("dblock_4_12", DBlock(4, 12), [torch.rand(2, 4, 16, 16) - 0.3], lambda m, x: m(x))
    torch.manual_seed(2)
    record("dblock_12_12_keep", DBlock(12, 12, keep_same_output=True), [torch.rand(2, 12, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(3)
    record("dblock_4_8_norelu", DBlock(4, 8, first_relu=False), [torch.rand(2, 4, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(4)
    record("dblock3d_4_8_norelu", DBlock(4, 8, conv_type="3d", first_relu=False), [torch.rand(2, 4, 4, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(5)
    record("dblock3d_8_16", DBlock(8, 16, conv_type="3d"), [torch.rand(2, 8, 6, 8, 8) - 0.3], lambda m, x: m(x))
    torch.manual_seed(6)
    record("dblock_4_12_eval", DBlock(4, 12), [torch.rand(2, 4, 16, 16) - 0.3], lambda m, x: m(x), train=False)
    # ---- GBlock / UpsampleGBlock (dgmr/common.py:17-155) ----
    torch.manual_seed(7)
    record("gblock_8_8", GBlock(8, 8), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(8)
    record("gblock_8_4", GBlock(8, 4), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(9)
    record("upgblock_8_4", UpsampleGBlock(8, 4), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x))
    torch.manual_seed(10)
    record("gblock_8_8_eval", GBlock(8, 8), [torch.randn(2, 8, 8, 8)], lambda m, x: m(x), train=False)
    # ---- LBlock / Attention (dgmr/common.py:241-300, dgmr/layers/Attention.py) ----
    torch.manual_seed(11)
    record("lblock_8_12", LBlock(8, 12), [torch.randn(1, 8, 4, 4)], lambda m, x: m(x))
    torch.manual_seed(12)
    att = AttentionLayer(16, 16)
    with torch.no_grad():
        att.gamma.fill_(0.7)
    record("attention_16", att, [torch.randn(2, 16, 4, 6)], lambda m, x: m(x))
    # ---- ConvGRU (dgmr/layers/ConvGRU.py) ----
    torch.manual_seed(13)
    record("convgru_8_4_T3", ConvGRU(8 + 4, 4, 3),
           [torch.randn(3, 2, 8, 8, 8), torch.randn(2, 4, 8, 8)], lambda m, xs, h: m(list(xs), h))
    # ---- Context / Latent stacks (dgmr/common.py:303-497) ----
    torch.manual_seed(14)
    record("context_128", ContextConditioningStack(1, 128), [torch.rand(2, 4, 1, 32, 32)], lambda m, x: m(x),
           grad_params=False)
    torch.manual_seed(15)
    lat = LatentConditioningStack((8, 2, 2), 288)
    with torch.no_grad():
        lat.att_block.gamma.fill_(0.5)
    torch.manual_seed(150)
    z = torch.distributions.normal.Normal(torch.Tensor([0.0]), torch.Tensor([1.0])).sample((8, 2, 2))
    z = torch.permute(z, (3, 0, 1, 2)).contiguous()

    def lat_call(m, zz):
        torch.manual_seed(150)  # the reference draws z itself (common.py:481-483); same seed -> same z
        return m(torch.zeros(1))

    record("latent_288", lat, [z], lat_call, grad_params=False)
    # ---- Sampler (dgmr/generators.py:20-182) ----
    torch.manual_seed(16)
    smp = Sampler(forecast_steps=2, latent_channels=64, context_channels=32)
    cond = [torch.randn(2, 4, 16, 16), torch.randn(2, 8, 8, 8), torch.randn(2, 16, 4, 4), torch.randn(2, 32, 2, 2)]
    record("sampler_64_32_T2", smp, cond + [torch.randn(1, 64, 2, 2)],
           lambda m, c0, c1, c2, c3, l: m([c0, c1, c2, c3], l), grad_params=False)
    # ---- Discriminators, reduced depth (dgmr/discriminators.py) ----
    torch.manual_seed(17)
    sd_ = SpatialDiscriminator(input_channels=1, num_timesteps=3, num_layers=1)
    xs = torch.rand(2, 6, 1, 16, 16)
    torch.manual_seed(170)
    idxs = torch.randint(low=0, high=6, size=(3,))

    def sdisc_call(m, x, _idxs):
        torch.manual_seed(170)
        return m(x)

    record("spatial_disc_L1", sd_, [xs, idxs], sdisc_call, grad_params=False)
    torch.manual_seed(18)
    td = TemporalDiscriminator(input_channels=1, num_layers=1)
    record("temporal_disc_L1", td, [torch.rand(2, 8, 1, 32, 32)], lambda m, x: m(x), grad_params=False)


if __name__ == "__main__":
    main()
