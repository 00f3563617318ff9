"""Pin the oracle (oracle/dgmr_oracle.py) against outputs of the UNMODIFIED reference (tests/golden/).

CPU only.  The fixtures were produced by oracle/gen_golden.py in the build container.
Tolerance: fp32 on both sides, same torch CPU kernels underneath -> 1e-5 relative to the tensor scale.
"""
import pytest
import torch

from conftest import load_golden, split_golden
from oracle import dgmr_oracle as O

TOL = 2e-5


def _close(a, b, tol=TOL, what=""):
    # error relative to the tensor's max magnitude, plus a small absolute floor (gradients that are
    # mathematically zero, e.g. a conv bias feeding BatchNorm, are ~1e-6 rounding noise on both sides)
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= tol * scale + 2e-5, f"{what}: abs err {err:.3e} at scale {scale:.3e}"


def _run(name, fn, float_param_grads=True):
    rec, meta = load_golden(name)
    sd, buf1, ins, outs, cots, grad_in, grad_p = split_golden(rec)
    train = meta["train"] == "True"
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("_u", "_v", "running_mean", "running_var")):
            v.requires_grad_(True)
    xin = [x.clone().requires_grad_(True) if x.is_floating_point() else x for x in ins]
    res = fn(sd, xin, train)
    res = list(res) if isinstance(res, (tuple, list)) else [res]
    assert len(res) == len(outs)
    for i, (r, o) in enumerate(zip(res, outs)):
        assert r.shape == o.shape
        _close(r.detach(), o, what=f"{name} out.{i}")
    loss = sum((r * c).sum() for r, c in zip(res, cots))
    loss.backward()
    for i, g in grad_in.items():
        _close(xin[i].grad, g, what=f"{name} grad.in.{i}")
    for k, g in grad_p.items():
        assert sd[k].grad is not None, k
        _close(sd[k].grad, g, tol=5e-5, what=f"{name} grad.p.{k}")
    for k, b in buf1.items():
        if b.is_floating_point():
            _close(sd[k].detach(), b, what=f"{name} buf1.{k}")
        else:
            assert torch.equal(sd[k], b), k


@pytest.mark.parametrize("name,kw", [
    ("dblock_4_12", {}),
    ("dblock_12_12_keep", {"keep_same_output": True}),
    ("dblock_4_8_norelu", {"first_relu": False}),
    ("dblock3d_4_8_norelu", {"first_relu": False}),
    ("dblock3d_8_16", {}),
    ("dblock_4_12_eval", {}),
])
def test_dblock(name, kw):
    _run(name, lambda sd, x, tr: O.dblock(sd, "", x[0], tr, **kw))


@pytest.mark.parametrize("name,up", [("gblock_8_8", False), ("gblock_8_4", False), ("upgblock_8_4", True),
                                     ("gblock_8_8_eval", False)])
def test_gblock(name, up):
    _run(name, lambda sd, x, tr: O.gblock(sd, "", x[0], tr, upsample=up))


def test_lblock():
    _run("lblock_8_12", lambda sd, x, tr: O.lblock(sd, "", x[0]))


def test_attention():
    _run("attention_32", lambda sd, x, tr: O.attention(sd, "", x[0]))


def test_convgru():
    _run("convgru_8_4_T3", lambda sd, x, tr: O.conv_gru(sd, "", list(x[0]), x[1], tr))


def test_context_stack():
    _run("context_128", lambda sd, x, tr: O.context_stack(sd, "", x[0], tr))


def test_latent_stack():
    _run("latent_256", lambda sd, x, tr: O.latent_stack(sd, "", x[0], tr))


def test_sampler():
    _run("sampler_64_32_T2", lambda sd, x, tr: O.sampler(sd, "", x[:4], x[4], 2, tr))


def test_spatial_discriminator():
    _run("spatial_disc_L1", lambda sd, x, tr: O.spatial_discriminator(sd, "", x[0], x[1].tolist(), tr))


def test_temporal_discriminator():
    _run("temporal_disc_L1", lambda sd, x, tr: O.temporal_discriminator(sd, "", x[0], tr))
