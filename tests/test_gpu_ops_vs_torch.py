"""Single operators of the HIP path against torch's own CPU ops in float64, forward AND backward, at the awkward shapes of the
paper-size discriminators: odd numbers of call groups (5 temporal frames), 40-image batches, 8x8 ... 2x2 maps with 384 ... 1536
channels (split-K territory), 1x1 / 3x3 / 3x3x3 kernels, ReLU-on-load, residuals, pooling with an odd depth.  The golden fixtures
are small by necessity; this is where a size- or divisibility-dependent slip in one kernel shows, isolated from the rest.

Spectral norm is emulated exactly: y_q = conv(pre(x_q), W / (u_q^T W v_q)) with constant u_q, v_q per call group q - autograd on
that expression IS the chain rule of torch's parametrization (torch/nn/utils/parametrizations.py:515-521) the kernels implement.
Exact f32 arithmetic; 2e-5 of each tensor's max magnitude (fp32 summation over up to 13 824 terms against float64).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, what, tol=2e-5):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= tol * scale + 1e-12, f"{what}: abs err {err:.3e} at scale {scale:.3e} ({err / max(scale, 1e-300):.2e} rel)"


# n, spatial dims, cin, cout, k, groups, pre_relu, residual
CONV_SHAPES = [
    (40, (4, 4), 384, 768, 3, 5, True, False),     # temporal D intermediate_dblocks.2.first_conv_3x3 (5 frames x 8 samples)
    (40, (4, 4), 768, 768, 3, 5, True, False),     # ... last_conv_3x3
    (40, (4, 4), 384, 768, 1, 5, False, False),    # ... conv_1x1
    (40, (8, 8), 192, 384, 3, 5, True, False),     # intermediate_dblocks.1
    (40, (8, 8), 384, 384, 3, 5, True, False),
    (40, (2, 2), 768, 768, 3, 5, True, True),      # d_last (keep_same_output: residual add)
    (64, (2, 2), 768, 768, 3, 8, True, True),      # spatial d6
    (24, (16, 16), 96, 192, 3, 3, True, False),
    (10, (16, 16), 48, 96, 1, 5, False, False),
    (6, (5, 16, 16), 48, 96, 3, 3, True, False),   # 3-D block, odd depth
    (4, (11, 32, 32), 4, 48, 3, 2, False, False),  # temporal d1.first_conv_3x3 (no ReLU on the frames)
]


@pytest.mark.parametrize("mode", ["f32", "bf16x6"])
@pytest.mark.parametrize("n,dims,cin,cout,k,groups,relu,res", CONV_SHAPES)
def test_sn_conv_fwd_bwd_vs_torch(n, dims, cin, cout, k, groups, relu, res, mode):
    """`bf16x6` (three bf16 planes per operand, six MFMAs per product) is held to the SAME 2e-5 as the exact-f32 kernels: it is the
    arithmetic the discriminator forward runs in inside bench.py's default mode."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    torch.manual_seed(hash((n, dims, cin, cout, k)) % 1000)
    nd = len(dims)
    ks = (k,) * nd
    x = torch.randn(n, cin, *dims, dtype=torch.float64)
    w = torch.randn(cout, cin, *ks, dtype=torch.float64) * (cin * k ** nd) ** -0.5
    b = torch.randn(cout, dtype=torch.float64)
    kk = cin * k ** nd
    u = F.normalize(torch.randn(groups, cout, dtype=torch.float64), dim=1)
    v = F.normalize(torch.randn(groups, kk, dtype=torch.float64), dim=1)
    r = torch.randn(n, cout, *dims, dtype=torch.float64) if res else None
    cot = torch.randn(n, cout, *dims, dtype=torch.float64)
    # ---- float64 reference ----
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    conv = F.conv2d if nd == 2 else F.conv3d
    ys, sig = [], []
    per = n // groups
    for q in range(groups):
        sigma = torch.dot(u[q], wr.flatten(1) @ v[q])
        sig.append(sigma.detach())
        xin = xr[q * per:(q + 1) * per]
        ys.append(conv(F.relu(xin) if relu else xin, wr / sigma, br, padding=k // 2))
    yref = torch.cat(ys, 0)
    if res:
        yref = yref + rr
    (yref * cot).sum().backward()
    # ---- HIP ----
    mf = torch.channels_last if nd == 2 else torch.channels_last_3d
    xd = x.float().to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    wd = torch.nn.Parameter(w.float().to(DEV).contiguous(memory_format=mf))
    bd = torch.nn.Parameter(b.float().to(DEV))
    rd = r.float().to(DEV).contiguous(memory_format=mf).requires_grad_(True) if res else None
    inv_sigma = (1.0 / torch.stack(sig)).float().to(DEV)
    sn = ops.SNCall(inv_sigma, u.float().to(DEV), v.float().to(DEV), groups)
    S.set_precision(mode)
    try:
        y = ops.conv(xd, wd, bd, inv_sigma, rd, ops.ConvSpec(pre_relu=relu, sn=sn))
        (y * cot.float().to(DEV)).sum().backward()
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    _close(y, yref, "forward")
    _close(xd.grad, xr.grad, "input gradient")
    _close(wd.grad, wr.grad, "weight gradient")
    _close(bd.grad, br.grad, "bias gradient")
    if res:
        _close(rd.grad, rr.grad, "residual gradient")


@pytest.mark.parametrize("shape,pd", [((40, 384, 8, 8), 1), ((40, 768, 4, 4), 1), ((6, 48, 11, 32, 32), 2), ((4, 96, 5, 16, 16), 2)])
def test_pool_add_vs_torch(shape, pd):
    from skillful_nowcasting_amd import ops

    torch.manual_seed(1)
    x = torch.randn(*shape, dtype=torch.float64)
    pool = (lambda t: F.avg_pool2d(t, 2)) if len(shape) == 4 else (lambda t: F.avg_pool3d(t, 2))
    xr = x.clone().requires_grad_(True)
    out_ref = pool(xr)
    add = torch.randn_like(out_ref)
    ar = add.clone().requires_grad_(True)
    cot = torch.randn_like(out_ref)
    ((out_ref + ar) * cot).sum().backward()
    mf = torch.channels_last if len(shape) == 4 else torch.channels_last_3d
    xd = x.float().to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    ad = add.float().to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    out = ops.avg_pool_add(xd, ad, pd)
    (out * cot.float().to(DEV)).sum().backward()
    _close(out, out_ref + ar, "pool + add")
    _close(xd.grad, xr.grad, "pool input gradient")  # odd depth: the dropped last plane must receive exactly zero
    _close(ad.grad, ar.grad, "addend gradient")


def test_frames_to_batch_and_heads_vs_torch():
    from skillful_nowcasting_amd import ops

    torch.manual_seed(2)
    n, c, t, h, w = 8, 96, 5, 16, 16
    x = torch.randn(n, c, t, h, w, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    ref = xr.permute(2, 0, 1, 3, 4).reshape(t * n, c, h, w)
    ref_h = F.relu(ref).sum(dim=(2, 3))
    cot = torch.randn(t * n, c, dtype=torch.float64)
    (ref_h * cot).sum().backward()
    xd = x.float().to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    rep = ops.frames_to_batch(xd)
    head = ops.relu_sum_hw(rep)
    (head * cot.float().to(DEV)).sum().backward()
    _close(rep, ref, "frames_to_batch")
    _close(head, ref_h, "relu_sum_hw")
    _close(xd.grad, xr.grad, "gradient through relu_sum_hw and frames_to_batch")


@pytest.mark.parametrize("groups,n,c", [(5, 8, 768), (8, 4, 1536), (30, 32, 768)])
def test_batchnorm1d_groups_vs_torch(groups, n, c):
    """The discriminator heads: BatchNorm1d with batch statistics per call group (train mode), running statistics updated once
    per group in order."""
    from skillful_nowcasting_amd.nn import BatchNorm1d

    torch.manual_seed(3)
    x = torch.randn(groups * n, c, dtype=torch.float64) * 3 + 1
    cot = torch.randn(groups * n, c, dtype=torch.float64)
    ref_bn = torch.nn.BatchNorm1d(c).double()
    with torch.no_grad():
        ref_bn.weight.uniform_(0.5, 1.5)
        ref_bn.bias.normal_()
    xr = x.clone().requires_grad_(True)
    yref = torch.cat([ref_bn(xr[q * n:(q + 1) * n]) for q in range(groups)], 0)
    (yref * cot).sum().backward()
    bn = BatchNorm1d(c)
    with torch.no_grad():
        bn.weight.copy_(ref_bn.weight.detach().float())
        bn.bias.copy_(ref_bn.bias.detach().float())
    bn = bn.to(DEV).train()
    xd = x.float().to(DEV).requires_grad_(True)
    y = bn(xd, groups=groups)
    (y * cot.float().to(DEV)).sum().backward()
    _close(y, yref, "batchnorm1d", 1e-5)
    _close(xd.grad, xr.grad, "batchnorm1d input gradient", 1e-4)
    _close(bn.weight.grad, ref_bn.weight.grad, "gamma gradient", 1e-5)
    _close(bn.running_mean, ref_bn.running_mean, "running mean", 1e-5)
    _close(bn.running_var, ref_bn.running_var, "running var", 1e-5)


# ------------------------------------------------------------------------------------------------------------------------------
# whole D-blocks at the paper discriminators' shapes, `calls` reference calls per launch, against the float64 oracle
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,hw,calls,per,keep", [
    (96, 192, 16, 5, 8, False),     # temporal intermediate_dblocks.0
    (192, 384, 8, 5, 8, False),     # .1
    (384, 768, 4, 5, 8, False),     # .2
    (768, 768, 2, 5, 8, True),      # d_last
    (192, 384, 8, 8, 8, False),     # spatial intermediate_dblocks.2
])
def test_dblock_calls_vs_oracle(cin, cout, hw, calls, per, keep):
    from oracle import dgmr_oracle as O
    from skillful_nowcasting_amd.common import DBlock

    torch.manual_seed(cin + hw)
    blk = DBlock(cin, cout, keep_same_output=keep)
    sd = {k: v.detach().clone().double() for k, v in blk.state_dict().items()}
    pk = O.param_keys(sd, "")
    for k in pk:
        sd[k].requires_grad_(True)
    n = calls * per
    x = torch.randn(n, cin, hw, hw, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    outs = [O.dblock(sd, "", xr[q * per:(q + 1) * per], True, keep_same_output=keep) for q in range(calls)]
    yref = torch.cat(outs, 0)
    cot = torch.randn_like(yref)
    (yref * cot).sum().backward()
    blk = blk.to(DEV).train()
    xd = x.float().to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = blk(xd, calls=calls)
    (y * cot.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    _close(y, yref, "block output", 5e-5)
    _close(xd.grad, xr.grad, "block input gradient", 5e-5)
    named = dict(blk.named_parameters())
    for k in pk:
        if sd[k].grad is not None:
            _close(named[k].grad, sd[k].grad, "grad " + k, 5e-5)
    for k, v in blk.state_dict().items():
        if k.endswith(("._u", "._v")):
            _close(v, sd[k], k, 5e-5)


# n, h, w, cin, cout, groups, upsample, residual ("", "same", "up"): the sampler's BatchNorm producers
STATS_SHAPES = [  # (the window kernels take a conv from 192 output tiles up)
    (32, 32, 32, 96, 96, 4, False, "same"),    # 96-channel tile, residual
    (16, 16, 16, 192, 192, 2, True, ""),       # upsampling conv, 32-wide output rows, 64-channel tail of 192 = 128 + 64
    (8, 32, 32, 48, 48, 2, True, "up"),        # 64-wide rows, 64-channel tile (48 used), low-resolution residual
    (72, 8, 8, 768, 768, 3, False, "same"),    # 8x8 maps: two images per tile, 128-channel tile
    (128, 64, 64, 96, 96, 16, False, ""),      # enough tiles for the 256-pixel kernel
    (32, 16, 16, 384, 384, 4, False, "same"),  # 16-wide maps
]


@pytest.mark.parametrize("mode", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,h,w,cin,cout,groups,up,res", STATS_SHAPES)
def test_conv_output_statistics(mode, n, h, w, cin, cout, groups, up, res):
    """dgmr_conv_args.stats_out: the per-tile (sum y, sum y^2) a window conv leaves for the next BatchNorm reduce to the same batch
    statistics as a pass over y, the output itself is bit-identical with and without them, and bn_prepare(partials=) produces the
    same normalisation as bn_prepare reading y."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    torch.manual_seed(n + h + cin)
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    mf = torch.channels_last
    x = (torch.randn(n, cin, h, w) + 0.3).to(DEV).contiguous(memory_format=mf)
    wt = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) * (cin * 9) ** -0.5).to(DEV).contiguous(memory_format=mf))
    b = torch.nn.Parameter(torch.randn(cout).to(DEV))
    rshape = {"": None, "same": (n, cout, ho, wo), "up": (n, cout, ho // 2, wo // 2)}[res]
    r = torch.randn(*rshape).to(DEV).contiguous(memory_format=mf) if rshape else None
    inv_sigma = (torch.rand(groups) + 0.5).to(DEV)
    kk = cin * 9
    sn = ops.SNCall(inv_sigma, torch.zeros(groups, cout, device=DEV), torch.zeros(groups, kk, device=DEV), groups)
    S.set_precision(mode)
    try:
        with torch.no_grad():
            spec = dict(pre_relu=True, sn=sn, upsample=up, residual_up=(res == "up"))
            y0 = ops.conv(x, wt, b, inv_sigma, r, ops.ConvSpec(**spec))
            y1, partials = ops.conv(x, wt, b, inv_sigma, r, ops.ConvSpec(want_stats=True, **spec))
            assert partials is not None, "the LDS-DMA window kernel should have taken this conv"
            assert torch.equal(y0, y1)
            rows = partials.shape[0]
            assert rows % groups == 0 and partials.shape[1:] == (2, cout)
            yd = y1.double().view(groups, n // groups, cout, ho * wo)
            want = torch.stack([yd.sum((1, 3)), (yd * yd).sum((1, 3))], 1)  # [groups, 2, C]
            got = partials.double().view(groups, rows // groups, 2, cout).sum(1)
            _close(got, want, "partial sums", 2e-6)
            # the BatchNorm that follows: same affine from the partials as from a pass over y
            g = torch.nn.Parameter(torch.rand(cout, device=DEV) + 0.5)
            be = torch.nn.Parameter(torch.randn(cout, device=DEV))
            outs = []
            for part in (None, partials):
                rm, rv, nb = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
                st = ops.bn_prepare(y1, g, be, rm, rv, nb, 1e-5, 0.1, True, groups, None, part)
                outs.append((st.a, st.b, st.mean, st.rstd, rm, rv))
            for got_t, ref_t, what in zip(outs[1], outs[0], ("a", "b", "mean", "rstd", "running_mean", "running_var")):
                _close(got_t, ref_t, f"bn {what}", 5e-6)
    finally:
        S.set_precision("f32")
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,hw,c,groups", [(12, 32, 48, 3), (8, 64, 48, 8), (6, 16, 64, 2), (4, 32, 24, 1)])
def test_sampler_head_vs_torch(n, hw, c, groups):
    """ops.HeadFn (dgmr_head_*): relu(BatchNorm(x)) -> spectrally-normalised 1x1 conv to 4 channels, per call group its own batch
    statistics and sigma - forward, input gradient (through the batch statistics), weight / bias / gamma / beta gradients and the
    running statistics against torch in float64."""
    from skillful_nowcasting_amd import ops

    torch.manual_seed(c + hw)
    x = torch.randn(n, c, hw, hw, dtype=torch.float64) * 1.5 + 0.4
    w = torch.randn(4, c, 1, 1, dtype=torch.float64) * c ** -0.5
    b = torch.randn(4, dtype=torch.float64)
    gam = torch.rand(c, dtype=torch.float64) + 0.5
    bet = torch.randn(c, dtype=torch.float64) * 0.3
    u = F.normalize(torch.randn(groups, 4, dtype=torch.float64), dim=1)
    v = F.normalize(torch.randn(groups, c, dtype=torch.float64), dim=1)
    cot = torch.randn(n, 4, hw, hw, dtype=torch.float64)
    per = n // groups
    # ---- float64 reference ----
    xr, wr, br, gr, ber = (t.clone().requires_grad_(True) for t in (x, w, b, gam, bet))
    ys, sig = [], []
    for q in range(groups):
        xq = xr[q * per:(q + 1) * per]
        hq = F.relu(F.batch_norm(xq, None, None, gr, ber, True, 0.1, 1e-5))
        sigma = torch.dot(u[q], wr.flatten(1) @ v[q])
        sig.append(sigma.detach())
        ys.append(F.conv2d(hq, wr / sigma, br))
    yref = torch.cat(ys, 0)
    (yref * cot).sum().backward()
    # ---- HIP ----
    mf = torch.channels_last
    xd = x.float().to(DEV).contiguous(memory_format=mf).requires_grad_(True)
    wd = torch.nn.Parameter(w.float().to(DEV).contiguous(memory_format=mf))
    bd = torch.nn.Parameter(b.float().to(DEV))
    gd = torch.nn.Parameter(gam.float().to(DEV))
    bed = torch.nn.Parameter(bet.float().to(DEV))
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    nb = torch.zeros((), dtype=torch.long, device=DEV)
    st = ops.bn_prepare(xd.detach(), gd, bed, rm, rv, nb, 1e-5, 0.1, True, groups)
    inv_sigma = (1.0 / torch.stack(sig)).float().to(DEV)
    sn = ops.SNCall(inv_sigma, u.float().to(DEV), v.float().to(DEV), groups)
    spec = ops.ConvSpec(bn=st, sn=sn)
    assert ops._head_applies(xd, wd, inv_sigma, None, spec)
    y = ops.conv(xd, wd, bd, inv_sigma, None, spec)
    (y * cot.float().to(DEV)).sum().backward()
    torch.cuda.synchronize()
    _close(y, yref, "forward", 1e-5)
    _close(xd.grad, xr.grad, "input gradient", 5e-5)
    _close(wd.grad, wr.grad, "weight gradient", 2e-5)
    _close(bd.grad, br.grad, "bias gradient", 2e-5)
    _close(gd.grad, gr.grad, "gamma gradient", 2e-5)
    _close(bed.grad, ber.grad, "beta gradient", 2e-5)
