"""GPU parity: the HIP path (through the C ABI) against the reference goldens and against the CPU oracle.

Every test here needs a real MI355X (-m gpu).  Tolerances: fp32 on both sides; the MFMA f32 path is an
exact fmaf chain, so per-block outputs are compared at 2e-5 of the tensor's max magnitude (+2e-5 absolute
for gradients that are mathematically zero), whole stacks at 1e-4, the north-star bound being 1e-3.
"""
import pytest
import torch

from conftest import load_golden, split_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _close(a, b, tol, what, floor=2e-5):
    a = a.detach().cpu().float()
    b = b.detach().cpu().float()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * scale + floor, f"{what}: abs err {err:.3e} at scale {scale:.3e} (tol {tol})"


def _run_golden(name, build, call, tol=2e-5, ptol=1e-4):
    """Load golden `name`, build the module, load its state, run fwd+bwd on the GPU, compare everything."""
    import skillful_nowcasting_amd  # noqa: F401

    rec, meta = load_golden(name)
    sd0, buf1, ins, outs, cots, grad_in, grad_p = split_golden(rec)
    train = meta["train"] == "True"
    mod = build()
    missing = mod.load_state_dict(sd0, strict=True)
    mod = mod.to(DEV).train(train)
    xin = [x.to(DEV).requires_grad_(True) if x.is_floating_point() else x for x in ins]
    res = call(mod, *xin)
    res = list(res) if isinstance(res, (tuple, list)) else [res]
    assert len(res) == len(outs)
    for i, (r, o) in enumerate(zip(res, outs)):
        _close(r, o, tol, f"{name} out.{i}")
    loss = sum((r * c.to(DEV)).sum() for r, c in zip(res, cots))
    loss.backward()
    torch.cuda.synchronize()
    for i, g in grad_in.items():
        _close(xin[i].grad, g, ptol, f"{name} grad.in.{i}")
    params = dict(mod.named_parameters())
    for k, g in grad_p.items():
        assert params[k].grad is not None, f"{name}: no grad for {k}"
        # absolute floor 1e-4: a conv bias in front of BatchNorm has an exactly-zero gradient, what both sides hold is rounding
        # noise of O(1e-5) whose value depends on the summation order (here: float atomics of the fused bias-gradient pass)
        _close(params[k].grad, g, ptol, f"{name} grad.p.{k}", floor=1e-4)
    sd1 = mod.state_dict()
    for k, b in buf1.items():
        if b.is_floating_point():
            _close(sd1[k], b, ptol, f"{name} buf1.{k}")
        else:
            assert torch.equal(sd1[k].cpu(), b), k


@pytest.mark.parametrize("name,args,kw", [
    ("dblock_4_12", (4, 12), {}),
    ("dblock_12_12_keep", (12, 12), {"keep_same_output": True}),
    ("dblock_4_8_norelu", (4, 8), {"first_relu": False}),
    ("dblock3d_4_8_norelu", (4, 8), {"conv_type": "3d", "first_relu": False}),
    ("dblock3d_8_16", (8, 16), {"conv_type": "3d"}),
    ("dblock_4_12_eval", (4, 12), {}),
])
def test_dblock(name, args, kw):
    from skillful_nowcasting_amd.common import DBlock

    _run_golden(name, lambda: DBlock(*args, **kw), lambda m, x: m(x))


@pytest.mark.parametrize("name,cls,args", [
    ("gblock_8_8", "GBlock", (8, 8)), ("gblock_8_4", "GBlock", (8, 4)), ("upgblock_8_4", "UpsampleGBlock", (8, 4)),
    ("gblock_8_8_eval", "GBlock", (8, 8)),
])
def test_gblock(name, cls, args):
    from skillful_nowcasting_amd import common

    _run_golden(name, lambda: getattr(common, cls)(*args), lambda m, x: m(x))


def test_lblock():
    from skillful_nowcasting_amd.common import LBlock

    _run_golden("lblock_8_12", lambda: LBlock(8, 12), lambda m, x: m(x))


def test_attention():
    from skillful_nowcasting_amd.layers import AttentionLayer

    _run_golden("attention_32", lambda: AttentionLayer(32, 32), lambda m, x: m(x))


def test_convgru():
    from skillful_nowcasting_amd.layers import ConvGRU

    _run_golden("convgru_8_4_T3", lambda: ConvGRU(12, 4, 3), lambda m, xs, h: m(list(xs.unbind(0)), h))


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 2e-4)])
def test_convgru_shared_input_equals_repeated_input(prec, tol):
    """The sampler's first ConvGRU gets ONE latent map for every sample and step (generators.py:146-149).  The x_shared path
    (x parts on one map, gate gradients summed over the samples before the x-part data / weight gradients) must equal the
    plain path fed with T*B copies: outputs, d latent, d h0 and every parameter gradient."""
    import copy

    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd.layers import ConvGRU

    T, B, cx, ch = 3, 4, 16, 8
    S.set_precision(prec)
    try:
        torch.manual_seed(7)
        a = ConvGRU(cx + ch, ch, 3).to(DEV)
        b = copy.deepcopy(a)
        lat = torch.randn(1, cx, 8, 8, device=DEV)
        h0 = torch.randn(B, ch, 8, 8, device=DEV)
        cot = torch.randn(T * B, ch, 8, 8, device=DEV)
        res = []
        for mod, shared in ((a, True), (b, False)):
            l, h = lat.clone().requires_grad_(True), h0.clone().requires_grad_(True)
            x = l if shared else ops.repeat_batch(l, T * B)
            out = mod.forward_batched(x, h, T, x_shared=shared)
            (out * cot).sum().backward()
            res.append((out, l.grad, h.grad, {k: p.grad for k, p in mod.named_parameters()}))
        torch.cuda.synchronize()
        (oa, la, ha, pa), (ob, lb, hb, pb) = res
        _close(oa, ob, tol, "out")
        _close(la, lb, tol, "d latent")
        _close(ha, hb, tol, "d h0")
        assert pa.keys() == pb.keys() and len(pa) == 6
        for k in pa:
            _close(pa[k], pb[k], tol, f"grad {k}")
    finally:
        S.set_precision("f32")


def test_context_stack():
    from skillful_nowcasting_amd import ContextConditioningStack

    _run_golden("context_128", lambda: ContextConditioningStack(1, 128), lambda m, x: m(x), tol=1e-4)


def test_latent_stack():
    from skillful_nowcasting_amd import LatentConditioningStack

    _run_golden("latent_256", lambda: LatentConditioningStack((8, 2, 2), 256), lambda m, z: m.forward_latent(z), tol=1e-4)


def test_sampler():
    from skillful_nowcasting_amd import Sampler

    _run_golden("sampler_64_32_T2", lambda: Sampler(forecast_steps=2, latent_channels=64, context_channels=32),
                lambda m, c0, c1, c2, c3, l: m([c0, c1, c2, c3], l), tol=1e-4, ptol=5e-4)


def test_spatial_discriminator():
    from skillful_nowcasting_amd import SpatialDiscriminator

    def call(m, x, idxs):
        torch.manual_seed(170)  # same CPU draw as the golden run (discriminators.py:199)
        return m(x)

    _run_golden("spatial_disc_L1", lambda: SpatialDiscriminator(input_channels=1, num_timesteps=3, num_layers=1), call,
                tol=1e-4, ptol=5e-4)


def test_temporal_discriminator():
    from skillful_nowcasting_amd import TemporalDiscriminator

    _run_golden("temporal_disc_L1", lambda: TemporalDiscriminator(input_channels=1, num_layers=1), lambda m, x: m(x),
                tol=1e-4, ptol=5e-4)


def test_cpu_input_fails_loudly():
    from skillful_nowcasting_amd.common import DBlock

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DBlock(4, 8)(torch.rand(1, 4, 8, 8))
