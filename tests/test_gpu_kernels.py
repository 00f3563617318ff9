"""GPU: kernel variants of one operator against each other, through the C ABI (dgmr_conv_tune selects the variant).

  * LDS-window 3x3 conv: the LDS-DMA kernel (conv_win_glds.h) and the register-staged one (conv_bf16.h) run the same MFMA sequence
    on the same operands - bit-identical outputs; the implicit-GEMM kernel differs by summation order only;
  * the two LDS-window weight gradients - the wave-specialised kernel (wgrad_ws.h: loader waves + matrix waves reading their
    fragments with ds_read_b64_tr_b16; the library's choice) and the one-role kernel of round 2 (wgrad_win.h) - vs the im2col weight
    gradient: same per-group partial sums and bias gradient in every bf16 mode (tolerance: product rounding of the mode x summation
    order; 2e-5 of the largest entry in bf16x3).
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture()
def tuned():
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd._lib import call

    S.set_precision("bf16x3")
    yield lambda *v: call("dgmr_conv_tune", *v)
    call("dgmr_conv_tune", -1, -1, -1, -1)
    S.set_precision("f32")


# n, h, w (output), cin, cout, upsample, batchnorm-on-load, residual
CONV_CASES = [
    (4, 64, 64, 192, 96, False, True, True),
    (2, 128, 128, 96, 96, True, True, False),
    (3, 32, 32, 64, 192, False, False, True),
    (4, 16, 16, 96, 96, False, True, False),
    (4, 8, 8, 96, 192, False, False, False),
    (3, 64, 64, 48, 48, False, True, False),     # Cin and Cout tails of the 32-channel chunk / 64-column tile
    (2, 32, 32, 40, 96, False, False, False),
    (2, 32, 32, 24, 200, True, True, True),
    (2, 64, 64, 96, 128, False, False, False),
    (3, 32, 32, 96, 48, False, True, True),      # 48 output channels: three 16-column blocks in mode 5
    (2, 64, 32, 72, 32, True, False, True),      # 32 output channels, upsample-on-load, ragged chunk
    (6, 16, 16, 48, 16, False, True, False),     # one 16-column block
]


@pytest.mark.parametrize("n,h,w,cin,cout,up,bn,res", CONV_CASES)
def test_window_conv_variants_agree(tuned, n, h, w, cin, cout, up, bn, res):
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(1)
    hin, win = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n * hin * win * cin, device=DEV)
    wt = torch.randn(cout * 9 * cin, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(n, device=DEV) + 0.5
    a = torch.rand(n * cin, device=DEV) + 0.5
    b = torch.randn(n * cin, device=DEV) * 0.1
    r = torch.randn(n * h * w * cout, device=DEV) if res else None
    wsp = torch.empty(2 * wt.numel(), device=DEV, dtype=torch.int16)
    call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * 9, cin, 0, 0, 2, 0, ops._stream())
    ys = {}
    # register-staged window, LDS-DMA window, the same with private per-wave weight slices (no barrier between taps; 128- and
    # 64-column tiles), with 256-pixel tiles (where eligible), implicit GEMM
    # 5: 16 x 16 MFMA blocks for <= 48 output channels (another block shape, i.e. another fp32 summation order: compared to 1e-6)
    for mode in (1, 3, 4, 2, 5, 0):
        tuned(-1, -1, mode, -1)
        y = torch.full((n * h * w * cout,), float("nan"), device=DEV)
        ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, 1, h, w, cin, cout, 1, 3, 3, upsample=up, pre_a=a if bn else None,
                         pre_b=b if bn else None, pre_group=1, scale_group=1, residual=r, w_split=wsp)
        torch.cuda.synchronize()
        ys[mode] = y
    assert not torch.isnan(ys[3]).any()
    assert torch.equal(ys[1], ys[3]), f"window kernels differ: max {float((ys[1] - ys[3]).abs().max()):.3e}"
    assert torch.equal(ys[2], ys[3]), f"256-pixel-tile window kernel differs: max {float((ys[2] - ys[3]).abs().max()):.3e}"
    assert torch.equal(ys[4], ys[3]), f"private-slice window kernel differs: max {float((ys[4] - ys[3]).abs().max()):.3e}"
    tol = 2e-6 * float(ys[0].abs().max())
    assert float((ys[3] - ys[0]).abs().max()) <= tol
    assert not torch.isnan(ys[5]).any()
    assert float((ys[5] - ys[0]).abs().max()) <= tol, f"16-column-block kernel: {float((ys[5] - ys[0]).abs().max()):.3e} vs {tol:.3e}"
    assert float((ys[5] - ys[3]).abs().max()) <= 0.5 * tol


# n, h, w (output), cin, cout, batchnorm-on-load, residual
PHASE_CASES = [
    (8, 64, 64, 96, 96, True, False),      # 32-wide input rows, 96-channel tile
    (16, 32, 32, 192, 192, True, False),   # 16-wide input maps, 128 + 64 columns
    (64, 16, 16, 384, 384, True, False),   # 8x8 input maps: two images per tile
    (4, 128, 128, 96, 48, True, False),    # 48 of 64 columns
    (40, 128, 128, 96, 96, True, False),   # enough tiles for the 256-pixel kernel
    (6, 64, 64, 40, 200, False, True),     # ragged chunk, ragged columns, residual
]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,h,w,cin,cout,bn,res", PHASE_CASES)
def test_upsample_phases_match_upsampled_conv(n, h, w, cin, cout, bn, res, prec):
    """dgmr_conv_args.w_phase: the nearest-2x upsample + 3x3 conv as four 2x2 convs on the low-resolution input gives the sums of the
    conv on the upsampled map (same kernel family, same arithmetic; the tap sums are rounded to fp32 before the bf16 split, and the
    products are added in another order: 3e-5 of the output's magnitude in bf16x3, where products carry 16 bits; 2e-2 in plain bf16,
    where SUMMED weights are rounded to 8 bits once instead of each tap separately)."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(2)
    x = torch.randn(n * (h // 2) * (w // 2) * cin, device=DEV)
    wt = torch.randn(cout * 9 * cin, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(n, device=DEV) + 0.5
    a = torch.rand(n * cin, device=DEV) + 0.5
    b = torch.randn(n * cin, device=DEV) * 0.1
    r = torch.randn(n * h * w * cout, device=DEV) if res else None
    S.set_precision(prec)
    try:
        wsp = torch.empty(2 * wt.numel(), device=DEV, dtype=torch.int16)
        call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * 9, cin, 0, 0, 2, 0, ops._stream())
        sums = torch.empty(16 * cout * cin, device=DEV)
        call("dgmr_upsample_phase_weights", wt.data_ptr(), sums.data_ptr(), cout, cin, ops._stream())
        # the tap sums themselves, against torch
        w4 = wt.view(cout, 3, 3, cin)
        rows = {0: ((0, 1), (1, 3)), 1: ((0, 2), (2, 3))}  # parity -> (tap range of offset 0, of offset 1)
        want = torch.stack([torch.stack([torch.stack([w4[:, rows[py][ai][0]:rows[py][ai][1], rows[px][bi][0]:rows[px][bi][1]].sum((1, 2))
                                                      for bi in (0, 1)], 1) for ai in (0, 1)], 1)
                            for py in (0, 1) for px in (0, 1)], 0)  # [4][Cout][2][2][Cin]
        assert torch.allclose(sums.view(4, cout, 2, 2, cin), want, rtol=0, atol=1e-6)
        wph = torch.empty(2 * sums.numel(), device=DEV, dtype=torch.int16)
        call("dgmr_split_weights", sums.data_ptr(), wph.data_ptr(), 16 * cout, cin, 0, 0, 2, 0, ops._stream())
        ys = []
        for ph in (None, wph):
            y = torch.full((n * h * w * cout,), float("nan"), device=DEV)
            part = ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, 1, h, w, cin, cout, 1, 3, 3, upsample=True,
                                    pre_a=a if bn else None, pre_b=b if bn else None, pre_group=1, scale_group=1, residual=r,
                                    w_split=wsp, w_phase=ph, want_stats=True)
            torch.cuda.synchronize()
            assert part is not None
            ys.append((y, part))
    finally:
        S.set_precision("f32")
    (y0, p0), (y1, p1) = ys
    assert not torch.isnan(y1).any()
    tol = 3e-5 if prec == "bf16x3" else 2e-2
    assert float((y1 - y0).abs().max()) <= tol * float(y0.abs().max())
    # fused output statistics: per sample (8x8 input maps: per pair of samples, a tile is two images), against the tensor the kernel wrote
    rows = p1.shape[0]
    g = n if rows >= 4 * n else n // 2
    assert rows % (4 * g) == 0
    yd = y1.double().view(g, -1, cout)
    want_s = torch.stack([yd.sum(1), (yd * yd).sum(1)], 1)
    got_s = p1.double().view(g, rows // g, 2, cout).sum(1)
    assert float((got_s - want_s).abs().max()) <= 2e-6 * float(want_s.abs().max())


@pytest.mark.parametrize("n,d,h,w,cin,cout,relu,res", [
    (2, 6, 32, 32, 48, 48, True, True),      # the temporal discriminator's first 3-D block shape (channel tails on both sides)
    (3, 4, 16, 16, 16, 96, False, False),
    (1, 5, 32, 64, 96, 128, True, False),
])
def test_window_conv3d_matches_implicit_gemm(tuned, n, d, h, w, cin, cout, relu, res):
    """3x3x3 convs go through the LDS-DMA window kernel plane by plane (zero planes beyond the volume)."""
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(3)
    x = torch.randn(n * d * h * w * cin, device=DEV)
    wt = torch.randn(cout * 27 * cin, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(n, device=DEV) + 0.5
    r = torch.randn(n * d * h * w * cout, device=DEV) if res else None
    wsp = torch.empty(2 * wt.numel(), device=DEV, dtype=torch.int16)
    call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * 27, cin, 0, 0, 2, 0, ops._stream())
    ys = {}
    for mode in (3, 0):
        tuned(-1, -1, mode, -1)
        y = torch.full((n * d * h * w * cout,), float("nan"), device=DEV)
        ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, d, h, w, cin, cout, 3, 3, 3, pre_relu=relu, scale_group=1, residual=r,
                         w_split=wsp)
        torch.cuda.synchronize()
        ys[mode] = y
    assert not torch.isnan(ys[3]).any()
    assert float((ys[3] - ys[0]).abs().max()) <= 2e-6 * float(ys[0].abs().max())


# n, h, w, cin, cout, upsample, batchnorm-on-load, call groups
WGRAD_CASES = [
    (2, 32, 32, 40, 96, False, True, 2),
    (2, 64, 64, 32, 48, True, True, 1),
    (3, 32, 64, 96, 192, False, False, 3),
    (2, 16, 16, 64, 96, False, True, 2),
    (3, 16, 16, 32, 64, True, False, 1),
    (6, 32, 32, 96, 128, False, True, 3),
    (5, 32, 32, 64, 96, False, True, 1),     # 80 tiles over 40 slabs ... an odd number of tiles per slab where the plan says so
    (1, 64, 32, 24, 200, True, False, 1),    # ragged input chunk (24 of 32 channels), 200 = 3 x 64 + 8 output channels
    (3, 16, 32, 96, 288, False, True, 3),    # 288 = 3 x 96 output-channel tiles
    (3, 32, 32, 96, 48, False, True, 3),     # 48 output channels: the pixel-split tile (wgrad_ws.h PSPLIT), three input-channel chunks
    (4, 16, 16, 48, 48, False, False, 2),    # ... on 16-wide maps, ragged second chunk (48 = 32 + 16 input channels)
    (5, 64, 32, 24, 40, False, True, 1),     # ... 40 of 48 columns, 24 of 32 input channels, an odd number of tiles per slab
]
# 3 x 3 x 3 convs (n, depth, h, w, cin, cout, call groups): the temporal discriminator's first blocks - one window launch per depth tap
WGRAD_CASES_3D = [(2, 6, 32, 32, 48, 96, 2), (1, 5, 16, 32, 8, 48, 1), (3, 2, 32, 64, 96, 96, 3)]


@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16x6", 2e-6), ("bf16", 6e-3)])
@pytest.mark.parametrize("n,d,h,w,cin,cout,groups", WGRAD_CASES_3D)
def test_window_wgrad_3d_matches_im2col_wgrad(tuned, n, d, h, w, cin, cout, groups, prec, tol):
    test_window_wgrad_matches_im2col_wgrad(tuned, n, h, w, cin, cout, False, False, groups, prec, tol, depth=d)


@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16x6", 2e-6), ("bf16", 6e-3)])
@pytest.mark.parametrize("n,h,w,cin,cout,up,bn,groups", WGRAD_CASES)
def test_window_wgrad_matches_im2col_wgrad(tuned, n, h, w, cin, cout, up, bn, groups, prec, tol, depth=1):
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import WgradArgs, call

    S.set_precision(prec)
    torch.manual_seed(2)
    hin, win = (h // 2, w // 2) if up else (h, w)
    kd = 3 if depth > 1 else 1
    x = torch.randn(n * depth * hin * win * cin, device=DEV)
    dy = torch.randn(n * depth * h * w * cout, device=DEV)
    a = torch.rand(groups * cin, device=DEV) + 0.5
    b = torch.randn(groups * cin, device=DEV) * 0.3
    k = 9 * kd * cin
    res = {}
    for mode in (0, 1, 2, 3):  # im2col, one-role window kernel (2-D only: im2col again for 3-D), wave-specialised kernel with 3 / 4 matrix waves
        tuned(-1, -1, -1, mode)
        bias = torch.zeros(cout, device=DEV)
        wa = WgradArgs()
        wa.x, wa.dy = x.data_ptr(), dy.data_ptr()
        wa.pre_a, wa.pre_b = (a.data_ptr(), b.data_ptr()) if bn else (None, None)
        wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = n, depth, h, w, cin, cout
        wa.KD, wa.KH, wa.KW = kd, 3, 3
        wa.upsample, wa.pre_relu, wa.pre_group, wa.groups = int(up), int(not bn), n // groups, groups
        wa.bias_grad = bias.data_ptr()
        call("dgmr_conv_wgrad_plan", ctypes.byref(wa))
        ns = wa.nsplit
        assert ns >= groups and ns % groups == 0
        partial = torch.full((ns, cout, k), float("nan"), device=DEV)
        wa.partial = partial.data_ptr()
        call("dgmr_conv_wgrad", ctypes.byref(wa), ops._stream())
        torch.cuda.synchronize()
        res[mode] = (partial.view(groups, ns // groups, cout, k).double().sum(1), bias.double())
    g0, b0 = res[0]
    for mode in (1, 2, 3):
        g1, b1 = res[mode]
        assert not torch.isnan(g1).any(), mode
        assert float((g0 - g1).abs().max()) <= tol * float(g0.abs().max()), mode
        assert float((b0 - b1).abs().max()) <= 2e-5 * float(b0.abs().max()), mode


@pytest.mark.parametrize("prec,tol", [("bf16x3", 3e-5), ("bf16x6", 3e-6)])
@pytest.mark.parametrize("cx,ch,hw,b,draws", [(96, 48, 64, 6, 1), (192, 96, 32, 12, 2), (384, 192, 16, 32, 1), (768, 384, 8, 96, 6)])
def test_convgru_fused_gates_match_separate_launches(prec, tol, cx, ch, hw, b, draws):
    """DGMR_EPI_GRU_GATES2: the read and update gate convs of a ConvGRU step as ONE launch with 2 C output columns (both convolve the
    same h, dgmr/layers/ConvGRU.py:69-76) against the three-launch step: layer outputs, input / initial-state gradients and every
    parameter gradient.  Same products, same K order; the 96- / 128-column tile of the fused launch groups them into other MFMA
    blocks than the separate launches' tiles, hence a rounding-level tolerance instead of bit equality."""
    import copy

    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd.layers import ConvGRU

    T = 3
    S.set_precision(prec)
    fused_launches = []
    orig_launch = ops._launch_conv

    def spy(*a, **k):
        r = orig_launch(*a, **k)
        if k.get("gates2") is not None:
            fused_launches.append(r is not NotImplemented)
        return r

    ops._launch_conv = spy
    keep = ops._GRU_FUSE_GATES
    try:
        torch.manual_seed(5)
        a = ConvGRU(cx + ch, ch, 3).to("cuda")
        bmod = copy.deepcopy(a)
        x = torch.randn(T * b, cx, hw, hw, device="cuda")
        h0 = torch.randn(b, ch, hw, hw, device="cuda")
        cot = torch.randn(T * b, ch, hw, hw, device="cuda")
        lay = ops.CallLayout(draws, T, time_major=True) if draws > 1 else None
        res = []
        for mod, fuse in ((a, True), (bmod, False)):
            ops._GRU_FUSE_GATES = fuse
            xs, hs = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
            out = mod.forward_batched(xs, hs, T, draws=draws, layout=lay)
            (out * cot).sum().backward()
            torch.cuda.synchronize()
            res.append((out.detach(), xs.grad, hs.grad, {k: p.grad.clone() for k, p in mod.named_parameters()},
                        {k: v.clone() for k, v in mod.state_dict().items() if k.endswith(("_u", "_v"))}))
    finally:
        ops._launch_conv = orig_launch
        ops._GRU_FUSE_GATES = keep
        S.set_precision("f32")
    assert fused_launches and all(fused_launches) and len(fused_launches) == T, fused_launches  # the fused path really ran, every step
    (o1, gx1, gh1, gp1, st1), (o2, gx2, gh2, gp2, st2) = res

    def close(u, v, what, t=tol):
        err = (u.double() - v.double()).abs().max().item() / max(v.double().abs().max().item(), 1e-30)
        assert err <= t, f"{what}: {err:.2e}"

    close(o1, o2, "outputs")
    close(gx1, gx2, "d x", 4 * tol)
    close(gh1, gh2, "d h0", 4 * tol)
    for k in gp2:
        close(gp1[k], gp2[k], "grad " + k, 10 * tol)
    for k in st2:
        assert torch.equal(st1[k], st2[k]), k  # spectral-norm state does not depend on how the convs are launched


@pytest.mark.parametrize("n,h,w,cin,cout,groups", [(4, 16, 16, 64, 96, 2), (6, 8, 8, 96, 192, 3), (2, 32, 32, 40, 48, 1)])
def test_upsampling_conv_weight_gradient_paths_agree(n, h, w, cin, cout, groups):
    """Weight / bias / scale gradient of an upsampling 3x3 conv (nearest-2x fused into the operand load, GBlock's first conv): the
    wave-specialised window kernel on the full-resolution map (the default where the output rows are 16 or 32 k pixels wide) against
    the pair-sum path (dgmr_upsample_wgrad_sums + a 1x1 weight gradient on the low-resolution map, kept for the other geometries;
    `ops._UP_WGRAD_SUMS` forces it) and against exact f32."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    torch.manual_seed(3)
    mf = torch.channels_last
    x = (torch.randn(n, cin, h, w) + 0.3).to(DEV).contiguous(memory_format=mf)
    w0 = (torch.randn(cout, cin, 3, 3) * (cin * 9) ** -0.5).to(DEV).contiguous(memory_format=mf)
    b0 = torch.randn(cout).to(DEV)
    inv_sigma = (torch.rand(groups) + 0.5).to(DEV)
    u = torch.nn.functional.normalize(torch.randn(groups, cout, device=DEV), dim=1)
    v = torch.nn.functional.normalize(torch.randn(groups, cin * 9, device=DEV), dim=1)
    gy = torch.randn(n, cout, 2 * h, 2 * w).to(DEV).contiguous(memory_format=mf)

    def grads(prec, sums):
        S.set_precision(prec)
        old = ops._UP_WGRAD_SUMS
        ops._UP_WGRAD_SUMS = sums
        try:
            wt, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
            xr = x.clone().requires_grad_(True)
            sn = ops.SNCall(inv_sigma, u, v, groups)
            y = ops.conv(xr, wt, b, inv_sigma, None, ops.ConvSpec(pre_relu=True, sn=sn, upsample=True))
            y.backward(gy)
            ops.join_side_streams()
            torch.cuda.synchronize()
            return [t.detach().clone() for t in (ops.grad_buffer(wt), ops.grad_buffer(b), xr.grad)]
        finally:
            ops._UP_WGRAD_SUMS = old
            S.set_precision("f32")

    ref = grads("f32", False)
    direct = grads("bf16x3", False)
    summed = grads("bf16x3", True)
    for got, what in ((direct, "window"), (summed, "pair sums")):
        for g, r, name in zip(got, ref, ("weight", "bias", "input")):
            err = float((g.double() - r.double()).abs().max()) / float(r.double().abs().max())
            assert err <= 1e-4, f"{what}: {name} gradient rel err {err:.2e}"


@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 6e-3)])
@pytest.mark.parametrize("n,h,w,cin,cout,groups", [(4, 16, 16, 64, 96, 2), (6, 32, 32, 96, 192, 3), (2, 64, 64, 40, 48, 1), (3, 32, 64, 96, 96, 1)])
def test_upsampling_conv_weight_gradient_by_phases(n, h, w, cin, cout, groups, prec, tol):
    """The same gradient with the wave-specialised kernel's PHASE mode (dgmr_conv_tune wgrad_window = 4): four 2 x 2-tap gradients on
    the low-resolution map, one per output-pixel parity, each written into the filter taps it feeds - against the 3 x 3-tap gradient
    on the upsampled map (same bf16 planes and products, another summation order) and against exact f32."""
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(4)
    mf = torch.channels_last
    x = (torch.randn(n, cin, h, w) + 0.3).to(DEV).contiguous(memory_format=mf)
    w0 = (torch.randn(cout, cin, 3, 3) * (cin * 9) ** -0.5).to(DEV).contiguous(memory_format=mf)
    b0 = torch.randn(cout).to(DEV)
    inv_sigma = (torch.rand(groups) + 0.5).to(DEV)
    u = torch.nn.functional.normalize(torch.randn(groups, cout, device=DEV), dim=1)
    v = torch.nn.functional.normalize(torch.randn(groups, cin * 9, device=DEV), dim=1)
    gy = torch.randn(n, cout, 2 * h, 2 * w).to(DEV).contiguous(memory_format=mf)

    def grads(mode, wgrad_window):
        S.set_precision(mode)
        call("dgmr_conv_tune", -1, -1, -1, wgrad_window)
        try:
            wt, b = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
            xr = x.clone().requires_grad_(True)
            sn = ops.SNCall(inv_sigma, u, v, groups)
            y = ops.conv(xr, wt, b, inv_sigma, None, ops.ConvSpec(pre_relu=True, sn=sn, upsample=True))
            y.backward(gy)
            ops.join_side_streams()
            torch.cuda.synchronize()
            return [t.detach().clone() for t in (ops.grad_buffer(wt), ops.grad_buffer(b))]
        finally:
            call("dgmr_conv_tune", -1, -1, -1, -1)
            S.set_precision("f32")

    ref = grads("f32", -1)
    direct = grads(prec, 3)
    phases = grads(prec, 4)
    for g, d, r, name in zip(phases, direct, ref, ("weight", "bias")):
        assert not torch.isnan(g).any(), name
        scale = float(r.double().abs().max())
        assert float((g.double() - d.double()).abs().max()) / scale <= tol, f"{name}: phases vs the 3 x 3 kernel"
        assert float((g.double() - r.double()).abs().max()) / scale <= (1e-4 if prec == "bf16x3" else 2e-2), f"{name}: phases vs exact f32"


# ------------------------------------------------------------------------------------------------------------------------------
# the wave-specialised persistent window kernel (conv_win_ws.h, dgmr_conv_tune window = 7) and the 16-column tile
# ------------------------------------------------------------------------------------------------------------------------------
def _ws_cases():
    import importlib.util
    import os

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("ws_check", os.path.join(ROOT, "tools", "ws_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_wave_specialised_window_conv_is_bit_identical_to_the_one_role_kernel(prec):
    """Every mode of conv3x3_ws_kernel (plain / phase / pooled; 96- and 128-column blocks; residual, half-resolution residual, relu
    mask with and without BatchNorm affine; ragged item counts; 8- and 16-pixel-wide maps; Cin and Cout tails) against the one-role
    LDS-DMA kernel with the same block shapes (window = 6): y bit for bit - the same MFMA sequence and the same epilogue
    expressions per element - and the fused BatchNorm partial sums, which the loaders add in another order, to 1e-6 after folding
    all rows in float64.  The library's own choice (window = -1: other tiles) differs by summation order only."""
    import skillful_nowcasting_amd as S

    ws = _ws_cases()
    S.set_precision(prec)
    try:
        for case in ws.SMALL:
            r = ws.run_case(case, prec)
            assert r["nan"] == 0 and r["exact"], (case[0], r)
            assert r["stats_rel"] < 1e-6, (case[0], r)
            assert r["vs_lib"] < (2e-5 if prec == "bf16x3" else 2e-2), (case[0], r)
    finally:
        S.set_precision("f32")


@pytest.mark.parametrize("n,d,h,w,cin,cout,kd", [(24, 1, 64, 64, 48, 4, 1), (4, 6, 32, 32, 48, 4, 3), (16, 1, 32, 32, 96, 8, 1)])
def test_thin_output_tile_matches_the_other_window_tiles(tuned, n, d, h, w, cin, cout, kd):
    """<= 8 output channels (the data gradients towards the discriminators' 4-channel inputs, 2-D and 3-D) through the 16-column window
    tile (the library's choice) against the 64-column tile (window = 3) and the implicit-GEMM kernel (window = 0)."""
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(5)
    x = torch.randn(n * d * h * w * cin, device=DEV)
    wt = torch.randn(cout * kd * 9 * cin, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    scale = torch.full((1,), 0.7, device=DEV)
    msk = torch.randn(n * d * h * w * cout, device=DEV)
    wsp = torch.empty(2 * wt.numel(), device=DEV, dtype=torch.int16)
    call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * kd * 9, cin, 0, 0, 2, 0, ops._stream())
    ys = []
    for win in (-1, 3, 0):
        tuned(-1, -1, win, -1)
        y = torch.full((n * d * h * w * cout,), float("nan"), device=DEV)
        ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, d, h, w, cin, cout, kd, 3, 3, mask_src=msk, w_split=wsp)
        torch.cuda.synchronize()
        ys.append(y)
    sc = ys[2].abs().max().item()
    assert not torch.isnan(ys[0]).any()
    assert (ys[0] - ys[1]).abs().max().item() <= 2e-6 * sc
    assert (ys[0] - ys[2]).abs().max().item() <= 5e-6 * sc


# Round 5: phase launches with ONE workgroup per ROW parity that computes both column parities from one staged halo (conv_win_glds.h
# PAIR; dgmr_debug_flags 256 / DGMR_PHASE_PAIR=1) against the default four-workgroups-per-tile scheme: the same MFMA sequence per output element - bit-identical
# outputs and fused statistics, with BatchNorm-on-load, residual at half resolution, 32- and 16-wide input maps, 96 ... 768 columns.
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,h,w,cin,cout,bn,res_up", [(8, 64, 64, 96, 96, True, False), (16, 32, 32, 192, 192, True, True), (40, 128, 128, 96, 96, True, True),
                                                      (12, 32, 32, 384, 384, False, False), (6, 64, 64, 48, 288, True, False)])
def test_phase_pair_kernel_is_bit_identical_to_one_workgroup_per_parity(n, h, w, cin, cout, bn, res_up, prec):
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(4)
    x = torch.randn(n * (h // 2) * (w // 2) * cin, device=DEV)
    wt = torch.randn(cout * 9 * cin, device=DEV) * 0.05
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(n, device=DEV) + 0.5
    a = torch.rand(n * cin, device=DEV) + 0.5
    b = torch.randn(n * cin, device=DEV) * 0.1
    r = torch.randn(n * (h // 2) * (w // 2) * cout, device=DEV) if res_up else None
    S.set_precision(prec)
    outs = []
    try:
        wsp = torch.empty(2 * wt.numel(), device=DEV, dtype=torch.int16)
        call("dgmr_split_weights", wt.data_ptr(), wsp.data_ptr(), cout * 9, cin, 0, 0, 2, 0, ops._stream())
        sums = torch.empty(16 * cout * cin, device=DEV)
        call("dgmr_upsample_phase_weights", wt.data_ptr(), sums.data_ptr(), cout, cin, ops._stream())
        wph = torch.empty(2 * sums.numel(), device=DEV, dtype=torch.int16)
        call("dgmr_split_weights", sums.data_ptr(), wph.data_ptr(), 16 * cout, cin, 0, 0, 2, 0, ops._stream())
        for flags in (0, 256):
            call("dgmr_debug_flags", flags)
            y = torch.full((n * h * w * cout,), float("nan"), device=DEV)
            part = ops._launch_conv(x, wt.data_ptr(), bias, scale, y, n, 1, h, w, cin, cout, 1, 3, 3, upsample=True,
                                    pre_a=a if bn else None, pre_b=b if bn else None, pre_group=1, scale_group=1, residual=r,
                                    residual_up=res_up, w_split=wsp, w_phase=wph, want_stats=True)
            torch.cuda.synchronize()
            assert part is not None and part is not NotImplemented
            outs.append((y, part.clone()))
    finally:
        call("dgmr_debug_flags", 0)
        S.set_precision("f32")
    (y0, p0), (y1, p1) = outs
    assert not torch.isnan(y1).any()
    assert torch.equal(y0, y1), f"outputs differ: {(y0 - y1).abs().max().item():.3e}"
    # statistics rows: [tiles x 4 phases][2][Cout]; the two schemes tile the map alike (128-pixel tiles) only where the four-workgroup scheme
    # does not pick its 256-pixel tile - compare the per-sample sums, which both must reproduce
    g = n
    s0 = p0.double().view(g, -1, 2, cout).sum(1)
    s1 = p1.double().view(g, -1, 2, cout).sum(1)
    assert torch.allclose(s0, s1, rtol=1e-6, atol=1e-6 * float(s0.abs().max()))


# conv + AvgPool (+ shortcut) as one operator (ops.ConvSpec.pool_out: DBlock's tail) against the conv followed by the pooling kernel.
# n, d, h, w (the conv's map), cin, cout, shortcut
POOL_OUT_CASES = [
    (24, 1, 64, 64, 96, 96, True),     # 96-column tile, 32-wide pooled map
    (24, 1, 64, 64, 48, 48, True),     # 16 x 16 blocks
    (96, 1, 32, 32, 192, 192, False),  # 16-wide pooled map
    (400, 1, 16, 16, 96, 128, True),   # 8 x 8 pooled map: two images per tile
    (2, 12, 64, 64, 48, 48, True),     # 3x3x3: spatial half in the conv, depth pair average behind it
    (5, 5, 64, 64, 48, 96, True),      # odd depth: the last plane is dropped like AvgPool3d does
    (2, 1, 8, 8, 96, 96, True),        # too small for the window kernel: conv + pooling kernel (the fallback must agree with itself)
]


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-4), ("bf16x6", 5e-6), ("bf16", 2e-2), ("f32", 2e-6)])
@pytest.mark.parametrize("n,d,h,w,cin,cout,res", POOL_OUT_CASES)
def test_conv_with_pooled_output_matches_conv_then_pool(n, d, h, w, cin, cout, res, prec, tol):
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    torch.manual_seed(3)
    is3d = d > 1
    pd = 2 if is3d else 1
    mf = torch.channels_last_3d if is3d else torch.channels_last
    xs = (n, cin, d, h, w) if is3d else (n, cin, h, w)
    ws = (cout, cin, 3, 3, 3) if is3d else (cout, cin, 3, 3)
    os_ = (n, cout, d // 2, h // 2, w // 2) if is3d else (n, cout, h // 2, w // 2)
    x0 = torch.randn(xs, device=DEV).contiguous(memory_format=mf)
    w0 = (torch.randn(ws, device=DEV) * 0.05).contiguous(memory_format=mf)
    b0 = torch.randn(cout, device=DEV)
    r0 = torch.randn(os_, device=DEV).contiguous(memory_format=mf) if res else None
    scale = torch.rand(1, device=DEV) + 0.5
    gy = torch.randn(os_, device=DEV).contiguous(memory_format=mf)
    launched = []
    orig = ops._launch_conv

    def spy(*a, **k):
        out = orig(*a, **k)
        if k.get("pool2"):
            launched.append(out is not NotImplemented)
        return out

    def run(fused):
        x, wt, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        for t in (wt, b):
            t.grad = torch.zeros_like(t)
        if fused:
            y = ops.conv(x, wt, b, scale, r, ops.ConvSpec(pre_relu=True, pool_out=True))
        else:
            y = ops.avg_pool_add(ops.conv(x, wt, b, scale, None, ops.ConvSpec(pre_relu=True)), r, pd)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), x.grad, wt.grad, b.grad, (r.grad if res else None)

    S.set_precision(prec)
    ops._launch_conv = spy
    try:
        got = run(True)
        ref = run(False)
    finally:
        ops._launch_conv = orig
        S.set_precision("f32")
    if prec != "f32" and h > 8:
        assert launched and all(launched), "the single-pass kernel was not taken"
    names = ("y", "dx", "dw", "dbias", "dresidual")
    for name, g, r_ in zip(names, got, ref):
        if g is None:
            assert r_ is None
            continue
        err = (g - r_).abs().max().item() / max(r_.abs().max().item(), 1e-30)
        # the backward pass runs the same kernels on the same dy either way: only y carries the different summation (the bias gradient's
        # per-workgroup sums meet in float atomics: 1e-6-level run-to-run differences)
        assert err <= (tol if name == "y" else 1e-5), f"{name}: {err:.2e}"


# Odd maps: DBlock's fused tail must floor like nn.AvgPool2d / AvgPool3d (96x96 inputs reach a 3x3 map: 6 -> 3 -> 1), forward and backward,
# against torch's own conv + avg_pool on the device (the fused tail raised on odd maps).
@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 2e-4)])
@pytest.mark.parametrize("n,d,h,w,cin,cout", [(6, 1, 3, 3, 96, 96), (4, 1, 5, 7, 48, 96), (3, 1, 33, 33, 48, 48), (2, 3, 5, 5, 48, 48), (2, 5, 9, 6, 48, 48)])
def test_pooled_conv_floors_odd_maps_like_avgpool(n, d, h, w, cin, cout, prec, tol):
    import torch.nn.functional as F

    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops

    torch.manual_seed(11)
    is3d = d > 1
    mf = torch.channels_last_3d if is3d else torch.channels_last
    xs = (n, cin, d, h, w) if is3d else (n, cin, h, w)
    ws = (cout, cin, 3, 3, 3) if is3d else (cout, cin, 3, 3)
    os_ = (n, cout, d // 2, h // 2, w // 2) if is3d else (n, cout, h // 2, w // 2)
    x0 = torch.randn(xs, device=DEV).contiguous(memory_format=mf)
    w0 = (torch.randn(ws, device=DEV) * 0.05).contiguous(memory_format=mf)
    b0 = torch.randn(cout, device=DEV)
    r0 = torch.randn(os_, device=DEV).contiguous(memory_format=mf)
    gy = torch.randn(os_, device=DEV).contiguous(memory_format=mf)
    scale = torch.ones(1, device=DEV)
    x, wt, b, r = (t.clone().requires_grad_(True) for t in (x0, w0, b0, r0))
    for t in (wt, b):
        t.grad = torch.zeros_like(t)
    S.set_precision(prec)
    try:
        y = ops.conv(x, wt, b, scale, r, ops.ConvSpec(pre_relu=True, pool_out=True))
        y.backward(gy)
        torch.cuda.synchronize()
    finally:
        S.set_precision("f32")
    xr, wr, br, rr = (t.double().clone().requires_grad_(True) for t in (x0, w0, b0, r0))
    conv = (F.conv3d if is3d else F.conv2d)(F.relu(xr), wr, br, padding=1)
    yr = (F.avg_pool3d if is3d else F.avg_pool2d)(conv, 2) + rr
    yr.backward(gy.double())
    assert tuple(y.shape) == tuple(yr.shape)
    for name, g, ref in (("y", y.detach(), yr.detach()), ("dx", x.grad, xr.grad), ("dw", wt.grad, wr.grad), ("db", b.grad, br.grad), ("dres", r.grad, rr.grad)):
        err = (g.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        assert err <= tol, f"{name}: {err:.2e}"


# The four-channel first-conv kernel (conv_stem4.h: exact fp32 on v_mfma_f32_16x16x4_f32, weights in registers) against the exact-f32
# implicit-GEMM kernel (forcing a tile variant through dgmr_conv_tune switches the special kernel off).
@pytest.mark.parametrize("n,d,h,w,cout,relu", [(6, 1, 64, 64, 96, False), (5, 1, 128, 128, 48, True), (2, 6, 64, 64, 48, False),
                                               (3, 3, 32, 32, 96, True), (1, 1, 8, 32, 48, False)])
@pytest.mark.parametrize("prec", ["f32", "bf16x6", "bf16x3"])
def test_four_channel_first_conv_kernel_matches_implicit_gemm(n, d, h, w, cout, relu, prec):
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(5)
    is3d = d > 1
    mf = torch.channels_last_3d if is3d else torch.channels_last
    x = torch.randn((n, 4, d, h, w) if is3d else (n, 4, h, w), device=DEV).contiguous(memory_format=mf)
    wt = (torch.randn((cout, 4, 3, 3, 3) if is3d else (cout, 4, 3, 3), device=DEV) * 0.2).contiguous(memory_format=mf)
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(1, device=DEV) + 0.5
    spec = ops.ConvSpec(pre_relu=relu)
    try:
        S.set_precision(prec)
        y = ops.conv(x, wt, bias, scale, None, spec)  # the library's choice: the stem kernel, in every mode
        S.set_precision("f32")
        call("dgmr_conv_tune", 1, -1, -1, -1)
        ref = ops.conv(x, wt, bias, scale, None, spec)
        torch.cuda.synchronize()
    finally:
        call("dgmr_conv_tune", -1, -1, -1, -1)
        S.set_precision("f32")
    assert not torch.isnan(y).any()
    err = (y - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err  # exact fp32 products on both sides, another summation order
    xr = torch.relu(x) if relu else x
    tref = (torch.nn.functional.conv3d if is3d else torch.nn.functional.conv2d)(xr.double(), wt.double(), None, padding=1) * scale.double() \
        + bias.double().view(1, -1, *([1] * (x.dim() - 2)))
    assert (y.double() - tref).abs().max().item() / tref.abs().max().item() < 2e-6


# The streaming 1x1-conv kernel (conv1x1.h) against the implicit-GEMM kernel (forcing a tile variant through dgmr_conv_tune switches the
# special kernel off): forward and data gradient, same bf16 planes and products, another summation order.
@pytest.mark.parametrize("n,h,w,cin,cout,relu", [(32, 64, 64, 96, 48, False), (128, 32, 32, 192, 96, True), (512, 16, 16, 384, 192, False),
                                                 (160, 32, 32, 80, 200, True), (40, 64, 64, 48, 384, False)])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 5e-3)])
def test_streaming_1x1_conv_matches_implicit_gemm(n, h, w, cin, cout, relu, prec, tol):
    import skillful_nowcasting_amd as S
    from skillful_nowcasting_amd import ops
    from skillful_nowcasting_amd._lib import call

    torch.manual_seed(9)
    x0 = torch.randn(n, cin, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    w0 = (torch.randn(cout, cin, 1, 1, device=DEV) * 0.1).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device=DEV)
    scale = torch.rand(1, device=DEV) + 0.5
    gy = torch.randn(n, cout, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    spec = ops.ConvSpec(pre_relu=relu)

    def run():
        x = x0.clone().requires_grad_(True)
        y = ops.conv(x, w0, bias, scale, None, spec)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), x.grad

    S.set_precision(prec)
    try:
        y, dx = run()
        call("dgmr_conv_tune", 2 if cout % 96 == 0 else (0 if cout > 64 else 3), -1, -1, -1)
        yr, dxr = run()
    finally:
        call("dgmr_conv_tune", -1, -1, -1, -1)
        S.set_precision("f32")
    for name, a, b in (("y", y, yr), ("dx", dx, dxr)):
        assert not torch.isnan(a).any(), name
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err <= tol, f"{name}: {err:.2e}"
    xr = torch.relu(x0) if relu else x0
    ref = torch.nn.functional.conv2d(xr.double(), w0.double()) * scale.double() + bias.double().view(1, -1, 1, 1)
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() <= (3e-4 if prec == "bf16x3" else 3e-2)
