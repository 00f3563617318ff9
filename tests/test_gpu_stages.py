"""Stage-by-stage parity of the generator forward against the oracle at the 512 x 512 stress configuration (BASELINE.json
configs[4]) and at the paper configuration: context stack (4 scales), latent stack, and inside the sampler every ConvGRU / 1x1 /
G-block / upsampling G-block output of the four levels.  A whole-model bound cannot say WHERE a size-dependent kernel path goes
wrong; this one does (every stage's error is listed in the failure message).  Exact f32 and bf16x3 arithmetic: 1e-3 of each stage's max.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_stages(O, sd, x, z, T):
    import torch.nn.functional as F

    st = {}
    cond = O.context_stack(sd, "conditioning_stack.", x, True)
    for i, c in enumerate(cond):
        st[f"context.{i}"] = c
    lat = O.latent_stack(sd, "latent_stack.", z.to(x.dtype), True)
    st["latent"] = lat
    p = "sampler."
    b = cond[0].shape[0]
    hs = [lat.repeat(b, 1, 1, 1)] * T
    names = [("convGRU1", "gru_conv_1x1", "g1", "up_g1"), ("convGRU2", "gru_conv_1x1_2", "g2", "up_g2"),
             ("convGRU3", "gru_conv_1x1_3", "g3", "up_g3"), ("convGRU4", "gru_conv_1x1_4", "g4", "up_g4")]
    for lvl, (gru, c11, g, upg) in enumerate(names):
        hs = list(O.conv_gru(sd, f"{p}{gru}.", hs, cond[3 - lvl], True))
        st[gru] = torch.cat(hs, 0)
        hs = [O.sn_conv(sd, f"{p}{c11}.", h, True) for h in hs]
        st[c11] = torch.cat(hs, 0)
        hs = [O.gblock(sd, f"{p}{g}.", h, True) for h in hs]
        st[g] = torch.cat(hs, 0)
        hs = [O.gblock(sd, f"{p}{upg}.", h, True, upsample=True) for h in hs]
        st[upg] = torch.cat(hs, 0)
    hs = [F.relu(O.batchnorm(sd, p + "bn.", h, True)) for h in hs]
    hs = [O.sn_conv(sd, p + "conv_1x1.", h, True) for h in hs]
    st["out"] = torch.stack([F.pixel_shuffle(h, 2) for h in hs], dim=1)
    return st


def _hip_stages(gen, x, z, T):
    from skillful_nowcasting_amd import ops

    st = {}
    cond = gen.conditioning_stack(x)
    for i, c in enumerate(cond):
        st[f"context.{i}"] = c
    lat = gen.latent_stack.forward_latent(z)
    st["latent"] = lat
    s = gen.sampler
    h = lat
    levels = (("convGRU1", "gru_conv_1x1", "g1", "up_g1"), ("convGRU2", "gru_conv_1x1_2", "g2", "up_g2"),
              ("convGRU3", "gru_conv_1x1_3", "g3", "up_g3"), ("convGRU4", "gru_conv_1x1_4", "g4", "up_g4"))
    for lvl, (gru, c11, g, upg) in enumerate(levels):
        h = getattr(s, gru).forward_batched(h, cond[3 - lvl], T, x_shared=(lvl == 0))
        st[gru] = h
        h = getattr(s, c11)(h, calls=T)
        st[c11] = h
        h = getattr(s, g)(h, calls=T)
        st[g] = h
        h = getattr(s, upg)(h, calls=T)
        st[upg] = h
    h = s.conv_1x1(h, bn=s.bn.prepare(h, T), calls=T)
    st["out"] = ops.d2s_frames(h, T)
    return st


@pytest.mark.parametrize("size,T,precision,tol", [(512, 18, "f32", 1e-3), (256, 18, "f32", 1e-3), (256, 18, "bf16x3", 1e-3),
                                                  (512, 18, "bf16x3", 1e-3), (256, 18, "bf16", 3e-1)])
def test_generator_stages_match_oracle(size, T, precision, tol):
    """`bf16` (operands rounded to 8 mantissa bits) is listed to show HOW its error grows through the stack (smoothly, ~1.3x per
    stage - precision, not a defect of one kernel path); its bound is what that growth reaches at the output."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    kw = dict(forecast_steps=T, output_shape=size, latent_channels=768, context_channels=384, generation_steps=6)
    torch.manual_seed(0)
    model = S.DGMR(**kw)
    with torch.no_grad():
        model.generator.latent_stack.att_block.gamma.fill_(0.3)
    sd = {k[len("generator."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("generator.")}
    model = model.to("cuda").train()
    x = torch.rand(1, 4, 1, size, size)
    torch.manual_seed(1)
    z = O.draw_latent((8, size // 32, size // 32))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = _oracle_stages(O, sd, x, z, T)
        S.set_precision(precision)
        try:
            got = _hip_stages(model.generator, x.cuda(), z.cuda(), T)
            torch.cuda.synchronize()
        finally:
            S.set_precision("f32")
    errs = {}
    for k, r in ref.items():
        g = got[k].detach().float().cpu()
        assert tuple(g.shape) == tuple(r.shape), (k, g.shape, r.shape)
        errs[k] = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
    table = "\n".join(f"  {k:18s} {e:.3e}" for k, e in errs.items())
    print(f"\nstage errors at {size}x{size} [{precision}]:\n{table}")
    bad = {k: e for k, e in errs.items() if not e <= tol}
    assert not bad, f"stages beyond {tol} at {size}x{size} [{precision}]:\n{table}"


# ------------------------------------------------------------------------------------------------------------------------------
# the discriminators' backward, block by block: output and GRADIENT AT THE OUTPUT of every D-block against the float64 oracle
# ------------------------------------------------------------------------------------------------------------------------------
def _oracle_temporal(O, sd, x):
    import torch.nn.functional as F

    p = "temporal_discriminator."
    st = {}

    def keep(name, t):
        t.retain_grad()
        st[name] = t
        return t

    h = F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))
    h = F.pixel_unshuffle(h, 2).permute(0, 2, 1, 3, 4)
    h = keep("d1", O.dblock(sd, p + "d1.", h, True, first_relu=False))
    h = keep("d2", O.dblock(sd, p + "d2.", h, True))
    h = h.permute(0, 2, 1, 3, 4)
    reps = []
    for t in range(h.shape[1]):
        rep = h[:, t]
        for i in range(3):
            rep = keep(f"intermediate_dblocks.{i}[{t}]", O.dblock(sd, f"{p}intermediate_dblocks.{i}.", rep, True))
        rep = keep(f"d_last[{t}]", O.dblock(sd, p + "d_last.", rep, True, keep_same_output=True))
        reps.append(O._d_head(sd, p, rep, True))
    return torch.sum(torch.stack(reps, dim=1), keepdim=True, dim=1), st


def test_temporal_discriminator_backward_stages():
    """Paper-size temporal discriminator on 8 sequences of 22 frames, exact f32, against the float64 oracle: the OUTPUT of d1, d2 (3-D
    blocks) and of every per-frame block at 1e-5 of its max magnitude; the GRADIENT arriving at each of those outputs and every
    parameter gradient in the l2 sense (<= 3e-3 / <= 3e-2) with the max-abs error and the worst elements listed.  Gradients cannot be
    held to a max-abs bound: a block output within 1e-7 of zero sits on the other side of the ReLU under a different fp32 summation
    order, that element's gradient toggles (the listing shows e.g. ONE element of 245 760 at half its reference value, every other
    one agreeing to 1e-6), and behind BatchNorm1d over 8 samples single elements carry 100x the typical gradient."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O
    from skillful_nowcasting_amd.common import DBlock

    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    model = S.DGMR(forecast_steps=18, output_shape=256, latent_channels=768, context_channels=384)
    sd = {k[len("discriminator."):]: v.detach().clone().double() for k, v in model.state_dict().items() if k.startswith("discriminator.")}
    for k in O.param_keys(sd, ""):
        sd[k].requires_grad_(True)
    torch.manual_seed(31)
    n = 8
    seq = torch.rand(n, 22, 1, 256, 256)
    cot = torch.randn(n, 1, 1)
    out_ref, st = _oracle_temporal(O, sd, seq.double())
    (out_ref * cot.double()).sum().backward()
    td = model.discriminator.temporal_discriminator.to("cuda").train()
    got = {}
    hooks = []

    def hook(name):
        def h(mod, inp, out):
            got[name + ".out"] = out.detach()
            out.register_hook(lambda g, nm=name: got.__setitem__(nm + ".dout", g.detach().clone()))
        return h

    for name, m in td.named_modules():
        if isinstance(m, DBlock):
            hooks.append(m.register_forward_hook(hook(name)))
    out = td(seq.cuda())
    (out * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    rows = []

    detail, l2s = [], {}

    def cmp(name, a, b):
        a, b = a.double().cpu(), b.double()
        assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
        scale = max(b.abs().max().item(), 1e-300)
        diff = (a - b).abs()
        rows.append((name, diff.max().item() / scale))
        l2 = (diff.pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item()
        l2s[name] = l2
        frac = (diff > 1e-3 * scale).double().mean().item()
        top = torch.topk(diff.flatten(), min(4, diff.numel())).indices
        pos = [tuple(int(v) for v in torch.unravel_index(i, a.shape)) for i in top]
        detail.append(f"  {name:34s} max {diff.max().item() / scale:.2e}  l2 {l2:.2e}  frac(>1e-3) {frac:.2e}  worst at "
                      + "; ".join(f"{q}: hip {a[q].item():+.4e} ref {b[q].item():+.4e}" for q in pos))

    cmp("scores", out.detach(), out_ref.detach())
    for name in ("d1", "d2"):  # oracle: [N, C, T, h, w]
        cmp(name + ".out", got[name + ".out"], st[name].detach())
        cmp(name + ".dout", got[name + ".dout"], st[name].grad)
    frames = st["d2"].shape[2]
    for name in ("intermediate_dblocks.0", "intermediate_dblocks.1", "intermediate_dblocks.2", "d_last"):
        ref_out = torch.cat([st[f"{name}[{t}]"].detach() for t in range(frames)], 0)  # frame-major, as the HIP batch
        ref_g = torch.cat([st[f"{name}[{t}]"].grad for t in range(frames)], 0)
        cmp(name + ".out", got[name + ".out"], ref_out)
        cmp(name + ".dout", got[name + ".dout"], ref_g)
    named = dict(td.named_parameters())
    for k in O.param_keys(sd, "temporal_discriminator."):
        if sd[k].grad is not None and sd[k].grad.abs().max().item() > 0:
            cmp("grad " + k[len("temporal_discriminator."):], named[k[len("temporal_discriminator."):]].grad, sd[k].grad)
    table = "\n".join(f"  {k:34s} {e:.3e}" for k, e in rows)
    print("\ntemporal discriminator, f32 vs float64 oracle:\n" + table + "\n" + "\n".join(detail))
    bad = [k for k, e in rows if (k.endswith(".out") or k == "scores") and not e <= (1e-3 if k == "scores" else 1e-5)]
    bad += [k for k, e in rows if k.endswith(".dout") and not l2s[k] <= 3e-3]
    bad += [k for k, e in rows if k.startswith("grad ") and not l2s[k] <= 3e-2]
    assert not bad, f"beyond the bounds: {bad}\n{table}"


def test_discriminators_on_160x160_inputs_floor_odd_maps_like_the_reference():
    """160x160 frames reach DBlocks with 5x5 maps (spatial: 160 -> 80 -> 40 -> 20 -> 10 -> 5 -> 2 -> 1; temporal: 6 frames -> 3 -> 1, maps
    40 -> 20 -> 10 -> 5 -> 2): the fused DBlock tail must floor like nn.AvgPool2d / AvgPool3d (dgmr/common.py:186-189), forward and
    backward, as the oracle does (96x96 is too small for the reference itself: its fifth D-block would pool a 1x1 map)."""
    import skillful_nowcasting_amd as S
    from oracle import dgmr_oracle as O

    torch.manual_seed(5)
    disc = S.Discriminator(input_channels=1)
    sd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    disc = disc.cuda().train()
    # eight samples: the heads' BatchNorm1d over a batch of two maps every feature to +-1 and its input gradient to rounding noise
    x = torch.rand(8, 6, 1, 160, 160)
    wgt = torch.randn(8, 2, 1)
    for prec, tol in (("f32", 1e-3), ("mixed", 1e-3)):
        disc.load_state_dict(sd)
        S.set_precision(prec)
        try:
            torch.manual_seed(9)
            xg = x.cuda().requires_grad_(True)
            out = disc(xg)
            (out * wgt.cuda()).sum().backward()
            torch.cuda.synchronize()
        finally:
            S.set_precision("f32")
        torch.manual_seed(9)
        idxs = torch.randint(0, 6, (8,)).tolist()
        xr = x.clone().requires_grad_(True)
        ref = O.discriminator({k: v.clone() for k, v in sd.items()}, "", xr, idxs, True)
        (ref * wgt).sum().backward()
        assert out.shape == ref.shape == (8, 2, 1)
        err = (out.detach().cpu() - ref.detach()).abs().max().item() / ref.detach().abs().max().item()
        assert err <= tol, f"{prec}: discriminator forward on odd maps {err:.2e}"
        gerr = (xg.grad.cpu() - xr.grad).abs().max().item() / xr.grad.abs().max().item()
        cos = torch.nn.functional.cosine_similarity(xg.grad.cpu().double().reshape(-1), xr.grad.double().reshape(-1), dim=0).item()
        # (relu kinks are not aligned here: a handful of flipped units on 2x2 / 1x1 maps move single gradient entries; direction and scale must agree)
        assert cos >= 0.999 and gerr <= 5e-2, f"{prec}: input gradient on odd maps: max err {gerr:.2e}, cosine {cos:.5f}"
