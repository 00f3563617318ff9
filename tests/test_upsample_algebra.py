"""The three decompositions of the upsampling conv (UpsampleGBlock.first_conv_3x3: nearest-2x, then 3x3; dgmr/common.py:142,148) that the
HIP path executes instead of the conv as written, restated in float64 on the CPU with the index formulas of the kernels
(csrc/conv.hip: phase_weights_kernel, pool2_weights_kernel; csrc/ops.hip: upsample_wgrad_sums_kernel) and checked against torch's own
conv / autograd.  Round 4 adds the weight gradient by output-pixel parity (csrc/wgrad_ws.h PHASE) and the DBlock tail - conv + AvgPool as one
pooled pass, 3-D plane by plane, shortcut conv on the pooled map (common.DBlock, ops.ConvFn._forward_pooled).  No GPU: this pins the algebra;
tests/test_gpu_kernels.py and tests/test_gpu_precision.py pin the kernels."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
N, CIN, COUT, H, W = 2, 5, 7, 6, 4  # low-resolution input H x W, output 2H x 2W


def _ref():
    x = torch.randn(N, CIN, H, W, dtype=torch.float64, requires_grad=True)
    w = torch.randn(COUT, CIN, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    dy = torch.randn_like(y)
    (y * dy).sum().backward()
    return x.detach(), w.detach(), y.detach(), dy, x.grad, w.grad


def _phase_taps(parity, a):
    """Filter taps (along one axis) that read input offset  parity - 1 + a  for output parity `parity`: phase_weights_kernel."""
    if parity == 0:
        return [0] if a == 0 else [1, 2]
    return [0, 1] if a == 0 else [2]


def test_forward_is_four_phase_convs():
    x, w, y, *_ = _ref()
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(y)
    for py in (0, 1):
        for px in (0, 1):
            acc = 0
            for a in (0, 1):
                for b in (0, 1):
                    wk = sum(w[:, :, ky, kx] for ky in _phase_taps(py, a) for kx in _phase_taps(px, b))  # [Cout, Cin] tap sum
                    patch = xp[:, :, py + a:py + a + H, px + b:px + b + W]  # input pixel (h + py - 1 + a, w + px - 1 + b)
                    acc = acc + torch.einsum("oc,nchw->nohw", wk, patch)
            out[:, :, py::2, px::2] = acc
    assert torch.allclose(out, y, rtol=0, atol=1e-12)


def test_data_gradient_is_one_pooled_pass_over_parity_planes():
    """dx = 2x2 sum pool of conv(dy, flipped w) = sum over the four pixel-parity planes of dy of a 2x2-tap conv: row u = 2a + 1 - p of the
    4x4 stride-2 kernel sums the flipped taps ky' with ky' + i = u, i in {0, 1} (pool2_weights_kernel); plane (p, q), tap (a, b) reads
    plane pixel (r + a - p, c + b - q)."""
    x, w, _, dy, dx_ref, _ = _ref()
    wf = w.flip(2, 3).permute(1, 0, 2, 3)  # [Cin, Cout, 3, 3]: dXup = conv(dy, wf)
    dx = torch.zeros_like(x)
    for p in (0, 1):
        for q in (0, 1):
            plane = F.pad(dy[:, :, p::2, q::2], (1, 1, 1, 1))  # [N, Cout, H + 2, W + 2], plane pixel (r, c) at [r + 1, c + 1]
            for a in (0, 1):
                for b in (0, 1):
                    u, v = 2 * a + 1 - p, 2 * b + 1 - q
                    wk = sum(wf[:, :, ky, kx] for ky in range(max(0, u - 1), min(2, u) + 1) for kx in range(max(0, v - 1), min(2, v) + 1))
                    patch = plane[:, :, 1 + a - p:1 + a - p + H, 1 + b - q:1 + b - q + W]
                    dx = dx + torch.einsum("io,nohw->nihw", wk, patch)
    assert torch.allclose(dx, dx_ref, rtol=0, atol=1e-12)


def test_weight_gradient_is_a_1x1_problem_on_pair_summed_planes():
    """z[n, r, c, co*9 + ky*3 + kx] = sum_{i,j in {0,1}} dy[n, co, 2r + 1 - ky + i, 2c + 1 - kx + j] (zero outside); then
    dW[co, ci, ky, kx] = sum_{n,r,c} z[...] * x[n, ci, r, c]; the centre plane's column sums are the bias gradient."""
    x, w, _, dy, _, dw_ref = _ref()
    dyp = F.pad(dy, (2, 2, 2, 2))  # index + 2
    z = torch.zeros(N, H, W, COUT, 3, 3, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            s = 0
            for i in (0, 1):
                for j in (0, 1):
                    r0, c0 = 1 - ky + i + 2, 1 - kx + j + 2
                    s = s + dyp[:, :, r0:r0 + 2 * H:2, c0:c0 + 2 * W:2]  # dy[2r + 1 - ky + i, 2c + 1 - kx + j]
            z[:, :, :, :, ky, kx] = s.permute(0, 2, 3, 1)
    dw = torch.einsum("nrcokl,nirc->oikl", z, x)
    assert torch.allclose(dw, dw_ref, rtol=0, atol=1e-11)
    assert torch.allclose(z[:, :, :, :, 1, 1].sum((0, 1, 2)), dy.sum((0, 2, 3)), rtol=0, atol=1e-11)


def test_weight_gradient_is_four_parity_gradients_on_the_low_resolution_map():
    """Round 4 (csrc/wgrad_ws.h PHASE): G[py, px][a, b] = sum_{n,r,c} dy[n, :, 2r + py, 2c + px] (x) x[n, :, r + py - 1 + a, c + px - 1 + b] -
    four 2x2-tap gradients with dy read at stride 2 - and filter tap (ky, kx) collects, from every parity, the window tap
    (a, b) = ((ky + 1 - py) >> 1, (kx + 1 - px) >> 1): each parity launch writes a complete 9-tap slab, the slabs add up to dW."""
    x, w, _, dy, _, dw_ref = _ref()
    xp = F.pad(x, (1, 1, 1, 1))
    dw = torch.zeros_like(w)
    for py in (0, 1):
        for px in (0, 1):
            dyp = dy[:, :, py::2, px::2]  # [N, Cout, H, W]: the parity's pixels
            g = {}
            for a in (0, 1):
                for b in (0, 1):
                    patch = xp[:, :, py + a:py + a + H, px + b:px + b + W]  # x[r + py - 1 + a, c + px - 1 + b]
                    g[a, b] = torch.einsum("nohw,nchw->oc", dyp, patch)
            slab = torch.zeros_like(w)  # what the launch of this parity writes
            for ky in range(3):
                for kx in range(3):
                    slab[:, :, ky, kx] = g[(ky + 1 - py) >> 1, (kx + 1 - px) >> 1]
            dw = dw + slab
    assert torch.allclose(dw, dw_ref, rtol=0, atol=1e-11)


def test_dblock_tail_is_one_pooled_pass_and_its_shortcut_commutes_with_the_pooling():
    """Round 4 (common.DBlock, ops.ConvFn._forward_pooled): AvgPool2d(2)(conv3x3(h)) = the 4x4 stride-2 conv whose taps are pool2_weights_kernel's
    sums of the UNflipped weight x 1/4, walked as the four pixel-parity planes of h; AvgPool3d(2)(conv3x3x3(h)) = the same per depth tap,
    then the depth pair average (a last odd plane dropped); and pool(conv1x1(x)) = conv1x1(pool(x))."""
    torch.manual_seed(1)
    h = torch.randn(N, CIN, 2 * H, 2 * W, dtype=torch.float64)
    w = torch.randn(COUT, CIN, 3, 3, dtype=torch.float64)
    b = torch.randn(COUT, dtype=torch.float64)
    ref = F.avg_pool2d(F.conv2d(h, w, b, padding=1), 2)

    def pooled_pass(hh, ww):  # hh: [N, Cin, 2H, 2W] -> [N, Cout, H, W], no bias
        out = 0
        for p in (0, 1):
            for q in (0, 1):
                plane = F.pad(hh[:, :, p::2, q::2], (1, 1, 1, 1))
                for a in (0, 1):
                    for bb in (0, 1):
                        u, v = 2 * a + 1 - p, 2 * bb + 1 - q  # row / column of the 4x4 kernel
                        wk = 0.25 * sum(ww[:, :, ky, kx] for ky in range(max(0, u - 1), min(2, u) + 1) for kx in range(max(0, v - 1), min(2, v) + 1))
                        patch = plane[:, :, 1 + a - p:1 + a - p + H, 1 + bb - q:1 + bb - q + W]
                        out = out + torch.einsum("oc,nchw->nohw", wk, patch)
        return out

    assert torch.allclose(pooled_pass(h, w) + b.view(1, -1, 1, 1), ref, rtol=0, atol=1e-12)
    # 3-D, odd depth: plane by plane over the three depth taps, then the depth pair average
    D = 5
    h3 = torch.randn(N, CIN, D, 2 * H, 2 * W, dtype=torch.float64)
    w3 = torch.randn(COUT, CIN, 3, 3, 3, dtype=torch.float64)
    ref3 = F.avg_pool3d(F.conv3d(h3, w3, b, padding=1), 2)
    h3p = F.pad(h3, (0, 0, 0, 0, 1, 1))
    sp = torch.stack([sum(pooled_pass(h3p[:, :, d + kd], w3[:, :, kd]) for kd in range(3)) for d in range(D)], dim=2)  # [N, Cout, D, H, W]
    got3 = 0.5 * (sp[:, :, 0:D - 1:2] + sp[:, :, 1:D:2]) + b.view(1, -1, 1, 1, 1)
    assert got3.shape == ref3.shape and torch.allclose(got3, ref3, rtol=0, atol=1e-12)
    # the shortcut
    w1 = torch.randn(COUT, CIN, 1, 1, dtype=torch.float64)
    assert torch.allclose(F.conv2d(F.avg_pool2d(h, 2), w1, b), F.avg_pool2d(F.conv2d(h, w1, b), 2), rtol=0, atol=1e-12)
    w13 = torch.randn(COUT, CIN, 1, 1, 1, dtype=torch.float64)
    assert torch.allclose(F.conv3d(F.avg_pool3d(h3, 2), w13, b), F.avg_pool3d(F.conv3d(h3, w13, b), 2), rtol=0, atol=1e-12)
