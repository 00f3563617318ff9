"""TEST INFRASTRUCTURE ONLY — functional CPU restatement of the reference DGMR hot path.

This file is the *oracle* for the parity tests: a stateless, functional re-expression of what
openclimatefix/skillful_nowcasting computes on `DGMR.training_step`,
written over a plain ``dict`` of tensors that uses the reference's ``state_dict`` key set.  It is
NOT a copy of the reference modules (there are no ``nn.Module`` classes here) and it is NOT part of
the product: nothing under ``skillful_nowcasting_amd/`` imports it, only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, and only as the checker.

Pinning: the reference ships **no numeric golden vectors** for this path (every
hot-path test asserts shapes / NaN-freeness only).  The oracle is therefore pinned against outputs of
the reference itself, run in the build container by ``oracle/gen_golden.py`` (which imports the
unmodified reference from /root/reference) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those fixtures.

Arithmetic that lives in third-party ``torch`` (unpinned in the reference's requirements.txt:1;
installed 2.10.0+rocm7.0) is restated from its published definition:
spectral norm  -> torch/nn/utils/parametrizations.py:454-521,
batch norm     -> torch.nn.BatchNorm{1,2}d train-mode definition (biased batch var, unbiased running var).

All functions are differentiable through torch autograd (CPU); ``sd`` values that should receive
gradients must have ``requires_grad=True``.  Buffers in ``sd`` (``_u``, ``_v``, ``running_mean``,
``running_var``, ``num_batches_tracked``) are updated IN PLACE exactly when the reference updates them.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

SN_EPS_DEFAULT = 1e-12  # torch spectral_norm default; used by DBlock / context / latent / sampler 1x1 / fc
SN_EPS_G = 1e-4  # dgmr/common.py:25,95 ; dgmr/layers/ConvGRU.py:17


# --------------------------------------------------------------------------------------
# third-party arithmetic (torch) restated
# --------------------------------------------------------------------------------------
def _l2_normalize(x: torch.Tensor, eps: float) -> torch.Tensor:
    # F.normalize(x, dim=0, eps): x / max(||x||_2, eps)
    return x / x.norm().clamp_min(eps)


def sn_weight(sd: SD, prefix: str, train: bool, eps: float = SN_EPS_DEFAULT) -> torch.Tensor:
    """W / sigma with one in-place power iteration per call in train mode.

    torch/nn/utils/parametrizations.py:454-521 (`_SpectralNorm._power_method`, `.forward`).
    """
    w = sd[prefix + "parametrizations.weight.original"]
    u = sd[prefix + "parametrizations.weight.0._u"]
    v = sd[prefix + "parametrizations.weight.0._v"]
    wm = w.flatten(1)
    if train:
        with torch.no_grad():
            u.copy_(_l2_normalize(torch.mv(wm, v), eps))
            v.copy_(_l2_normalize(torch.mv(wm.t(), u), eps))
    uc = u.clone()
    vc = v.clone()
    sigma = torch.dot(uc, torch.mv(wm, vc))
    return w / sigma


def sn_conv(sd: SD, prefix: str, x: torch.Tensor, train: bool, eps: float = SN_EPS_DEFAULT) -> torch.Tensor:
    """spectral_norm(ConvNd(k in {1,3}, stride 1, 'same' zero padding)) applied to x (NCHW / NCDHW)."""
    w = sn_weight(sd, prefix, train, eps)
    b = sd[prefix + "bias"]
    pad = w.shape[-1] // 2
    if w.ndim == 5:
        return F.conv3d(x, w, b, padding=pad)
    return F.conv2d(x, w, b, padding=pad)


def plain_conv(sd: SD, prefix: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    w = sd[prefix + "weight"]
    b = sd[prefix + "bias"] if bias else None
    return F.conv2d(x, w, b, padding=w.shape[-1] // 2)


def batchnorm(sd: SD, prefix: str, x: torch.Tensor, train: bool, eps: float = 1e-5, momentum: float = 0.1):
    """BatchNorm over every dim except channel dim 1 (BatchNorm2d on NCHW, BatchNorm1d on [N, C])."""
    g = sd[prefix + "weight"]
    b = sd[prefix + "bias"]
    rm = sd[prefix + "running_mean"]
    rv = sd[prefix + "running_var"]
    dims = [d for d in range(x.ndim) if d != 1]
    shape = [1, -1] + [1] * (x.ndim - 2)
    if train:
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            rm.mul_(1 - momentum).add_(momentum * mean.detach())
            rv.mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
            sd[prefix + "num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    xhat = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps)
    return xhat * g.view(shape) + b.view(shape)


# --------------------------------------------------------------------------------------
# blocks — dgmr/common.py
# --------------------------------------------------------------------------------------
# Tests only: RELU_HOOK(x, tag) replaces relu at the discriminator's kinks (tag = "<block prefix>in" / "<block prefix>mid" /
# "<discriminator prefix>head").  tests/conftest.py::KinkAligner uses it to evaluate the oracle on the SAME linear piece the
# implementation under test was on (x * mask with the implementation's own mask): a pre-activation within fp32 rounding of zero
# otherwise puts two correct implementations on different pieces of a piecewise-linear function, and behind the heads' BatchNorm1d
# that one element moves every gradient below it.  None (the default) = plain relu, i.e. the reference's arithmetic.
RELU_HOOK = None


def _relu(x: torch.Tensor, tag: str) -> torch.Tensor:
    return F.relu(x) if RELU_HOOK is None else RELU_HOOK(x, tag)


def dblock(sd: SD, p: str, x: torch.Tensor, train: bool, first_relu: bool = True, keep_same_output: bool = False):
    """dgmr/common.py:220-238 (DBlock.forward), 2-D and "3d" variants (AvgPool2d/3d k=2)."""
    w = sd[p + "first_conv_3x3.parametrizations.weight.original"]
    cout, cin = w.shape[0], w.shape[1]
    pool = (lambda t: F.avg_pool3d(t, 2, 2)) if w.ndim == 5 else (lambda t: F.avg_pool2d(t, 2, 2))
    if cin != cout:
        x1 = sn_conv(sd, p + "conv_1x1.", x, train)
        if not keep_same_output:
            x1 = pool(x1)
    else:
        x1 = x
    h = _relu(x, p + "in") if first_relu else x
    h = sn_conv(sd, p + "first_conv_3x3.", h, train)
    h = _relu(h, p + "mid")
    h = sn_conv(sd, p + "last_conv_3x3.", h, train)
    if not keep_same_output:
        h = pool(h)
    return x1 + h


def gblock(sd: SD, p: str, x: torch.Tensor, train: bool, upsample: bool = False):
    """dgmr/common.py:68-84 (GBlock.forward) and :139-155 (UpsampleGBlock.forward)."""
    cout = sd[p + "last_conv_3x3.parametrizations.weight.original"].shape[0]
    up = (lambda t: F.interpolate(t, scale_factor=2, mode="nearest")) if upsample else (lambda t: t)
    if upsample:
        sc = sn_conv(sd, p + "conv_1x1.", up(x), train, SN_EPS_G)
    elif x.shape[1] != cout:
        sc = sn_conv(sd, p + "conv_1x1.", x, train, SN_EPS_G)
    else:
        sc = x
    h = F.relu(batchnorm(sd, p + "bn1.", x, train))
    h = up(h)
    h = sn_conv(sd, p + "first_conv_3x3.", h, train, SN_EPS_G)
    h = F.relu(batchnorm(sd, p + "bn2.", h, train))
    h = sn_conv(sd, p + "last_conv_3x3.", h, train, SN_EPS_G)
    return h + sc


def lblock(sd: SD, p: str, x: torch.Tensor):
    """dgmr/common.py:288-300 (LBlock.forward) — plain convs, no spectral norm."""
    cout, cin = sd[p + "first_conv_3x3.weight"].shape[:2]
    if cin < cout:
        sc = torch.cat([x, plain_conv(sd, p + "conv_1x1.", x)], dim=1)
    else:
        sc = x
    h = plain_conv(sd, p + "first_conv_3x3.", F.relu(x))
    h = plain_conv(sd, p + "last_conv_3x3.", F.relu(h))
    return h + sc


def attention(sd: SD, p: str, x: torch.Tensor):
    """dgmr/layers/Attention.py:9-20,71-85.

    NB the reference feeds `query[b]` of shape [c, h, w] to an einsum written for [h, w, c]; i.e. the
    softmax runs over L = c*h "positions" with feature length w.  Restated literally.
    """
    q = plain_conv(sd, p + "query.", x, bias=False)
    k = plain_conv(sd, p + "key.", x, bias=False)
    v = plain_conv(sd, p + "value.", x, bias=False)
    outs = []
    for b in range(x.shape[0]):
        qb, kb, vb = q[b], k[b], v[b]  # [d0, d1, d2]
        k2 = kb.reshape(-1, kb.shape[-1])
        v2 = vb.reshape(-1, vb.shape[-1])
        beta = torch.softmax(torch.einsum("hwc,Lc->hwL", qb, k2), dim=-1)
        outs.append(torch.einsum("hwL,Lc->hwc", beta, v2))
    out = torch.stack(outs, dim=0)
    out = sd[p + "gamma"] * plain_conv(sd, p + "last_conv.", out, bias=False)
    return out + x


def conv_gru_cell(sd: SD, p: str, x: torch.Tensor, h: torch.Tensor, train: bool):
    """dgmr/layers/ConvGRU.py:57-85."""
    xh = torch.cat([x, h], dim=1)
    r = torch.sigmoid(sn_conv(sd, p + "read_gate_conv.", xh, train, SN_EPS_G))
    u = torch.sigmoid(sn_conv(sd, p + "update_gate_conv.", xh, train, SN_EPS_G))
    gated = torch.cat([x, r * h], dim=1)
    c = F.relu(sn_conv(sd, p + "output_conv.", gated, train, SN_EPS_G))
    out = u * h + (1.0 - u) * c
    return out, out


def conv_gru(sd: SD, p: str, xs: Sequence[torch.Tensor], h: torch.Tensor, train: bool):
    """dgmr/layers/ConvGRU.py:102-111."""
    outs = []
    for x in xs:
        o, h = conv_gru_cell(sd, p + "cell.", x, h, train)
        outs.append(o)
    return torch.stack(outs, dim=0)


# --------------------------------------------------------------------------------------
# stacks — dgmr/common.py, dgmr/generators.py, dgmr/discriminators.py
# --------------------------------------------------------------------------------------
def context_stack(sd: SD, p: str, x: torch.Tensor, train: bool):
    """dgmr/common.py:388-424 (ContextConditioningStack.forward, _mixing_layer)."""
    x = F.pixel_unshuffle(x, 2)  # [B, T, 4C, H/2, W/2]
    scales: List[List[torch.Tensor]] = [[], [], [], []]
    for i in range(x.shape[1]):
        s = x[:, i]
        for lvl in range(4):
            s = dblock(sd, f"{p}d{lvl + 1}.", s, train)
            scales[lvl].append(s)
    outs = []
    for lvl in range(4):
        st = torch.stack(scales[lvl], dim=1)  # b t c h w
        b, t, c, h, w = st.shape
        st = st.permute(0, 2, 1, 3, 4).reshape(b, c * t, h, w)  # "b t c h w -> b (c t) h w"
        outs.append(F.relu(sn_conv(sd, f"{p}conv{lvl + 1}.", st, train)))
    return tuple(outs)


def latent_stack(sd: SD, p: str, z: torch.Tensor, train: bool, use_attention: bool = True):
    """dgmr/common.py:469-497; `z` is the [1, 8, h, w] draw (the reference draws it on the CPU RNG, :481-483)."""
    z = sn_conv(sd, p + "conv_3x3.", z, train)
    z = lblock(sd, p + "l_block1.", z)
    z = lblock(sd, p + "l_block2.", z)
    z = lblock(sd, p + "l_block3.", z)
    if use_attention:
        z = attention(sd, p + "att_block.", z)
    z = lblock(sd, p + "l_block4.", z)
    return z


def draw_latent(shape: Sequence[int]) -> torch.Tensor:
    """The reference's z draw: Normal(0,1).sample(shape) -> [*shape, 1] -> permute(3,0,1,2) (common.py:481-483)."""
    dist = torch.distributions.normal.Normal(loc=torch.Tensor([0.0]), scale=torch.Tensor([1.0]))
    z = dist.sample(tuple(shape))
    return torch.permute(z, (3, 0, 1, 2))


def sampler(sd: SD, p: str, cond: Sequence[torch.Tensor], latent: torch.Tensor, forecast_steps: int, train: bool):
    """dgmr/generators.py:125-182 (Sampler.forward)."""
    b = cond[0].shape[0]
    latent = latent.repeat(b, 1, 1, 1)  # einops "b c h w -> (repeat b) c h w"
    hs: List[torch.Tensor] = [latent] * forecast_steps
    names = [("convGRU1", "gru_conv_1x1", "g1", "up_g1"), ("convGRU2", "gru_conv_1x1_2", "g2", "up_g2"),
             ("convGRU3", "gru_conv_1x1_3", "g3", "up_g3"), ("convGRU4", "gru_conv_1x1_4", "g4", "up_g4")]
    for lvl, (gru, c11, g, upg) in enumerate(names):
        hs = list(conv_gru(sd, f"{p}{gru}.", hs, cond[3 - lvl], train))
        hs = [sn_conv(sd, f"{p}{c11}.", h, train) for h in hs]
        hs = [gblock(sd, f"{p}{g}.", h, train) for h in hs]
        hs = [gblock(sd, f"{p}{upg}.", h, train, upsample=True) for h in hs]
    hs = [F.relu(batchnorm(sd, p + "bn.", h, train)) for h in hs]
    hs = [sn_conv(sd, p + "conv_1x1.", h, train) for h in hs]
    hs = [F.pixel_shuffle(h, 2) for h in hs]
    return torch.stack(hs, dim=1)


def generator(sd: SD, p: str, x: torch.Tensor, z: torch.Tensor, forecast_steps: int, train: bool):
    """dgmr/generators.py:207-212 with prefixes conditioning_stack./latent_stack./sampler. under `p`."""
    cond = context_stack(sd, p + "conditioning_stack.", x, train)
    lat = latent_stack(sd, p + "latent_stack.", z.to(x.dtype), train)
    return sampler(sd, p + "sampler.", cond, lat, forecast_steps, train)


def _count_children(sd: SD, prefix: str) -> int:
    return len({k[len(prefix):].split(".")[0] for k in sd if k.startswith(prefix)})


def _d_head(sd: SD, p: str, rep: torch.Tensor, train: bool):
    rep = torch.sum(_relu(rep, p + "head"), dim=[2, 3])
    rep = batchnorm(sd, p + "bn.", rep, train)
    w = sn_weight(sd, p + "fc.", train)
    return F.linear(rep, w, sd[p + "fc.bias"])


def spatial_discriminator(sd: SD, p: str, x: torch.Tensor, idxs: Sequence[int], train: bool):
    """dgmr/discriminators.py:196-232; `idxs` = the torch.randint(0, S, (8,)) draw (:199)."""
    n_mid = _count_children(sd, p + "intermediate_dblocks.")
    reps = []
    for idx in idxs:
        rep = F.avg_pool2d(x[:, int(idx)], 2)
        rep = F.pixel_unshuffle(rep, 2)
        rep = dblock(sd, p + "d1.", rep, train, first_relu=False)
        for i in range(n_mid):
            rep = dblock(sd, f"{p}intermediate_dblocks.{i}.", rep, train)
        rep = dblock(sd, p + "d6.", rep, train, keep_same_output=True)
        reps.append(_d_head(sd, p, rep, train))
    return torch.sum(torch.stack(reps, dim=1), keepdim=True, dim=1)


def temporal_discriminator(sd: SD, p: str, x: torch.Tensor, train: bool):
    """dgmr/discriminators.py:104-138."""
    n_mid = _count_children(sd, p + "intermediate_dblocks.")
    x = F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))
    x = F.pixel_unshuffle(x, 2)
    x = x.permute(0, 2, 1, 3, 4)
    x = dblock(sd, p + "d1.", x, train, first_relu=False)
    x = dblock(sd, p + "d2.", x, train)
    x = x.permute(0, 2, 1, 3, 4)
    reps = []
    for t in range(x.shape[1]):
        rep = x[:, t]
        for i in range(n_mid):
            rep = dblock(sd, f"{p}intermediate_dblocks.{i}.", rep, train)
        rep = dblock(sd, p + "d_last.", rep, train, keep_same_output=True)
        reps.append(_d_head(sd, p, rep, train))
    return torch.sum(torch.stack(reps, dim=1), keepdim=True, dim=1)


def discriminator(sd: SD, p: str, x: torch.Tensor, idxs: Sequence[int], train: bool):
    """dgmr/discriminators.py:39-44."""
    s = spatial_discriminator(sd, p + "spatial_discriminator.", x, idxs, train)
    t = temporal_discriminator(sd, p + "temporal_discriminator.", x, train)
    return torch.cat([s, t], dim=1)


# --------------------------------------------------------------------------------------
# losses — dgmr/losses.py:172-192,307-319 ; dgmr/dgmr.py:20-33
# --------------------------------------------------------------------------------------
def loss_hinge_disc(score_generated, score_real):
    return torch.mean(F.relu(1.0 - score_real)) + torch.mean(F.relu(1.0 + score_generated))


def loss_hinge_gen(score_generated):
    return -torch.mean(score_generated)


def grid_cell_loss(generated_mean, targets, precip_weight_cap: float = 24.0):
    """losses.py:172-192 with dgmr.py:33: weights = max(y+1, cap); returns ||d*w||_1 / T * H * W (sic)."""
    w = torch.clamp_min(targets + 1, precip_weight_cap)
    diff = ((generated_mean - targets) * w).abs().sum()
    return diff / targets.size(1) * targets.size(3) * targets.size(4)


# --------------------------------------------------------------------------------------
# Adam — torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0), single-tensor definition
# --------------------------------------------------------------------------------------
def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
              beta1: float, beta2: float, eps: float = 1e-8):
    """In-place Adam update (dgmr/dgmr.py:292-300 constructs two of these)."""
    with torch.no_grad():
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------------------
# the whole step — dgmr/dgmr.py:137-218
# --------------------------------------------------------------------------------------
_BUFFER_SUFFIXES = ("._u", "._v", "running_mean", "running_var", "num_batches_tracked")


def param_keys(sd: SD, prefix: str) -> List[str]:
    return [k for k in sd if k.startswith(prefix) and not k.endswith(_BUFFER_SUFFIXES)]


def training_step(sd: SD, images: torch.Tensor, future: torch.Tensor, hp: dict, opt: dict, capture: Optional[dict] = None):
    """One `DGMR.training_step` on the state dict `sd` (keys under ``generator.`` and ``discriminator.``).

    Restates dgmr/dgmr.py:137-218 literally, including activation checkpointing of the generator
    (:150,176), the un-detached predictions in the D pass (Q6), the extra forward at :213 (Q8) and the
    CPU-RNG draws (latent z: common.py:481-483; frame indices: discriminators.py:199).
    `hp`: forecast_steps, generation_steps, grid_lambda, gen_lr, disc_lr, beta1, beta2, precip_weight_cap,
    latent_shape, num_spatial_frames.  `opt`: {"step": {key: int}, "m": {key: t}, "v": {key: t}} (Adam state).
    Returns (d_loss, g_loss, grid_loss) as floats; `sd` and `opt` are updated in place.
    `capture` (tests): receives "backward_losses" (the three losses `manual_backward` is called on, dgmr.py:163,196), "d_grads" (one
    dict per discriminator pass: the gradients `d_opt.step()` consumes, :165) and "g_grads" (those of `g_opt.step()`, :200).
    `sd` may be float64 (the latent draw is cast to the images' dtype; the RNG stream is the same).
    """
    from torch.utils.checkpoint import checkpoint

    gp, dp = param_keys(sd, "generator."), param_keys(sd, "discriminator.")
    for k in gp + dp:
        sd[k].requires_grad_(True)
    T = hp["forecast_steps"]

    def gen(x):
        z = draw_latent(hp["latent_shape"]).to(x.dtype)
        return generator(sd, "generator.", x, z, T, True)

    def disc(x):
        idxs = torch.randint(low=0, high=x.size(1), size=(hp.get("num_spatial_frames", 8),))
        return discriminator(sd, "discriminator.", x, idxs.tolist(), True)

    if capture is not None:
        capture.update(backward_losses=[], d_grads=[], g_grads={})

    def adam(keys, lr):
        for k in keys:
            p = sd[k]
            if p.grad is None:
                continue
            if k not in opt["m"]:
                opt["m"][k] = torch.zeros_like(p)
                opt["v"][k] = torch.zeros_like(p)
                opt["step"][k] = 0
            opt["step"][k] += 1
            adam_step(p, p.grad, opt["m"][k], opt["v"][k], opt["step"][k], lr, hp["beta1"], hp["beta2"])

    real_sequence = torch.cat([images, future], dim=1)
    b = images.shape[0]
    for _ in range(2):
        for k in dp:
            sd[k].grad = None
        predictions = checkpoint(gen, images, use_reentrant=False)
        generated_sequence = torch.cat([images, predictions], dim=1)
        out = disc(torch.cat([real_sequence, generated_sequence], dim=0))
        s_real, s_gen = out[:b], out[b:]
        d_loss = loss_hinge_disc(s_gen[:, 0:1], s_real[:, 0:1]) + loss_hinge_disc(s_gen[:, 1:2], s_real[:, 1:2])
        d_loss.backward()
        if capture is not None:
            capture["backward_losses"].append(float(d_loss.detach()))
            capture["d_grads"].append({k: sd[k].grad.detach().clone() for k in dp if sd[k].grad is not None})
        adam(dp, hp["disc_lr"])
    predictions = [checkpoint(gen, images, use_reentrant=False) for _ in range(hp["generation_steps"])]
    gen_mean = torch.stack(predictions, dim=0).mean(dim=0)
    grid = grid_cell_loss(gen_mean, future, hp["precip_weight_cap"])
    scores = []
    for p_ in predictions:
        out = disc(torch.cat([real_sequence, torch.cat([images, p_], dim=1)], dim=0))
        scores.append(out[b:])
    g_loss = loss_hinge_gen(torch.cat(scores, dim=0)) + hp["grid_lambda"] * grid
    for k in gp:
        sd[k].grad = None
    g_loss.backward()
    if capture is not None:
        capture["backward_losses"].append(float(g_loss.detach()))
        capture["g_grads"] = {k: sd[k].grad.detach().clone() for k in gp if sd[k].grad is not None}
    adam(gp, hp["gen_lr"])
    gen(images)  # the logging forward (dgmr.py:213): advances u/v, BN statistics and the CPU RNG
    return float(d_loss.detach()), float(g_loss.detach()), float(grid.detach())


def validation_step(sd: SD, images: torch.Tensor, future: torch.Tensor, hp: dict, train: bool = False):
    """`DGMR.validation_step` (dgmr/dgmr.py:220-290) on the state dict: two (generator forward + discriminator loss) rounds, the
    generator loss over `generation_steps` draws, one more forward - no optimisation.  `train` is the module's mode (Lightning
    validates in eval mode; in train mode the forwards advance u / v and the BatchNorm statistics in `sd`).  Returns
    (d_loss, g_loss, grid_loss) as floats.  RNG consumption as in the reference: z per generator forward, frame indices per
    discriminator call."""
    T = hp["forecast_steps"]

    def gen(x):
        z = draw_latent(hp["latent_shape"])
        return generator(sd, "generator.", x, z, T, train)

    def disc(x):
        idxs = torch.randint(low=0, high=x.size(1), size=(hp.get("num_spatial_frames", 8),))
        return discriminator(sd, "discriminator.", x, idxs.tolist(), train)

    with torch.no_grad():
        real_sequence = torch.cat([images, future], dim=1)
        b = images.shape[0]
        for _ in range(2):
            predictions = gen(images)
            out = disc(torch.cat([real_sequence, torch.cat([images, predictions], dim=1)], dim=0))
            s_real, s_gen = out[:b], out[b:]
            d_loss = loss_hinge_disc(s_gen[:, 0:1], s_real[:, 0:1]) + loss_hinge_disc(s_gen[:, 1:2], s_real[:, 1:2])
        predictions = [gen(images) for _ in range(hp["generation_steps"])]
        grid = grid_cell_loss(torch.stack(predictions, dim=0).mean(dim=0), future, hp["precip_weight_cap"])
        scores = []
        for p_ in predictions:
            out = disc(torch.cat([real_sequence, torch.cat([images, p_], dim=1)], dim=0))
            scores.append(out[b:])
        g_loss = loss_hinge_gen(torch.cat(scores, dim=0)) + hp["grid_lambda"] * grid
        gen(images)
    return float(d_loss), float(g_loss), float(grid)
