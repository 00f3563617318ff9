"""Host-side operators of the DGMR step: thin ``torch.autograd.Function`` wrappers over libdgmr_hip.so.

PyTorch is plumbing here (device memory, streams, the autograd tape); every arithmetic op on the hot
path is a HIP kernel reached through the C ABI in ``include/dgmr_hip.h``.  There is no CPU or eager-torch
fallback: tensors that are not on a HIP device raise ``RuntimeError``.

Layout: activations are logical NCHW / NCDHW tensors held in ``channels_last`` / ``channels_last_3d``
memory format, i.e. physically N[D]HWC; conv weights are OIHW parameters held channels_last, i.e.
physically O[D]HWI — exactly what the kernels index.

Parameter gradients are accumulated by the kernels directly into ``param.grad`` (allocated zeroed on
first touch); the Functions return ``None`` for parameter inputs.  That is what lets the weight-gradient
epilogue fuse the spectral-norm chain rule and skips autograd's separate accumulate kernels.
"""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import ConvArgs, WgradArgs, call

from . import _core, _streams
from ._core import (BNState, CallLayout, SNCall, SPLITK_WS_BYTES, _copy, _dims, _p, _scratch, _splitk_ws, _stream, bn_prepare,  # noqa: F401
                    bias_rows, bump_weights_epoch, call_slots, deterministic, dot_buffer, empty_cl, grad_buffer, require_hip,
                    require_weight_layout, set_deterministic, set_grad_touch_hook, sums_buffer, colsum_tmp, to_cl, upload, weights_epoch)
from ._head_ops import (AttentionFn, AxpbyFn, BatchNorm1dFn, GridCellFn, HingeDiscFn, MeanFn, ReluSumHWFn, SNLinear1Fn, adam_update,  # noqa: F401
                        attention, axpby, relu_sum_hw)
from ._layout_ops import (CatChannelsFn, D2SFramesFn, FramesS2DFn, FramesToBatchFn, PoolAddFn, RepeatBatchFn, StackBatchFn,  # noqa: F401
                          SumGroupsFn, TimeToChannelsFn, UnstackBatchFn, avg_pool_add, cat_channels, d2s_frames, frames_s2d,
                          frames_to_batch, repeat_batch, stack_batch, sum_groups, time_to_channels, unstack_batch)
from ._streams import (_on_side_stream, branch_stream, defer_side_join, defer_wgrads, flush_deferred, join_side_streams,  # noqa: F401
                       side_streams)


PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2, "bf16x6": 3}
_PLANES = {0: 0, 1: 2, 2: 2, 3: 3}  # bf16 planes per pre-split weight tensor in each mode (dgmr_split_weights)
# "mixed": the generator and every backward pass in bf16x3, the discriminator's FORWARD in bf16x6.  The discriminator heads put a
# BatchNorm1d over the batch in front of the last linear layer (dgmr/discriminators.py:102,129,194,218); on a batch of similar
# sequences (iid-noise frames: bench.py's synthetic data) it divides by a spread ~1e-3 of the features and amplifies the forward's
# rounding error ~1e3-fold into every discriminator gradient: 2^-16 products (bf16x3) then cost 10-25 % of the gradient, fp32-grade
# products keep it at the reference's own fp32 level (tests/test_gpu_fullsize.py, iid-noise case).  Backward kernels are not
# amplified that way (the BatchNorm1d backward itself is exact fp32 VALU arithmetic on the forward's features).
_ALIASES = {"mixed": ("bf16x3", "bf16x6")}
_D_FORWARD_CODE = None  # precision code of the discriminator forward, None: the global mode


def _set_code(code: int):
    global _PRECISION_CODE
    call("dgmr_set_precision", code)
    _PRECISION_CODE = code


def set_precision(mode: str, discriminator_forward: Optional[str] = None):
    """Arithmetic of the conv contractions (tensors stay fp32 in HBM, accumulation is fp32):
    "f32"     exact fp32 MFMA (parity mode, the library default)
    "bf16x6"  three bf16 planes per operand, six MFMAs per product: fp32-faithful products (<= 2^-25 dropped) at 2.6x the f32 ceiling
    "bf16x3"  two planes, three MFMAs: products carry 16 significant bits (NOT fp32 arithmetic; forward within 1e-3 of fp32)
    "bf16"    operands rounded to bf16
    "mixed"   = set_precision("bf16x3", discriminator_forward="bf16x6"): bench.py's default (see _ALIASES above).
    `discriminator_forward`: a second mode for the forward pass of `Discriminator` only."""
    global _D_FORWARD_CODE
    if mode in _ALIASES:
        if discriminator_forward is not None:
            raise ValueError(f"precision {mode!r} already names its discriminator-forward mode")
        mode, discriminator_forward = _ALIASES[mode]
    if mode not in PRECISIONS or (discriminator_forward is not None and discriminator_forward not in PRECISIONS):
        raise ValueError(f"precision must be one of {sorted(PRECISIONS) + sorted(_ALIASES)}, got {mode!r} / {discriminator_forward!r}")
    _set_code(PRECISIONS[mode])
    d = None if discriminator_forward is None else PRECISIONS[discriminator_forward]
    _D_FORWARD_CODE = None if d == PRECISIONS[mode] else d


def get_precision() -> str:
    from ._lib import load

    code = int(load().dgmr_get_precision())
    name = next(k for k, v in PRECISIONS.items() if v == code)
    if _D_FORWARD_CODE is not None:
        pair = (name, next(k for k, v in PRECISIONS.items() if v == _D_FORWARD_CODE))
        return next((k for k, v in _ALIASES.items() if v == pair), f"{pair[0]}+d:{pair[1]}")
    return name


class discriminator_forward_precision:
    """Inside: launches are issued in the discriminator-forward mode of set_precision (no-op when none is set).  The mode is
    read by the library when a launch is ISSUED, so kernels already queued are unaffected; autograd runs the backward later,
    outside this scope, in the global mode."""

    def __enter__(self):
        self.prev = None
        if _D_FORWARD_CODE is not None and _D_FORWARD_CODE != _PRECISION_CODE:
            self.prev = _PRECISION_CODE
            _set_code(_D_FORWARD_CODE)

    def __exit__(self, *exc):
        if self.prev is not None:
            _set_code(self.prev)


# Weight gradients off the critical path: the backward chain only needs each conv's DATA gradient; its weight gradient (window /
# im2col kernel, slab reduce, spectral-norm finalize - latency-bound kernels at 20-30 % matrix-pipe occupancy) runs on a second
# stream beside the data-gradient convs of the layers below.  The main stream joins it when the backward pass ends
# (autograd engine callback), i.e. before anything can read a .grad.
_WGRAD_STREAM = __import__("os").environ.get("DGMR_WGRAD_STREAM", "1") != "0"

_GRU_KEEP_ALWAYS = bool(int(__import__("os").environ.get("DGMR_GRU_KEEP_ALWAYS", "0")))  # measurement switch


def spectral_sigma(w: torch.Tensor, u: torch.Tensor, v: torch.Tensor, scratch: torch.Tensor, eps: float, train: bool) -> SNCall:
    """One power iteration (train) + 1/sigma; u, v updated in place like the reference does."""
    require_hip(w, "spectral-norm weight")
    cout = w.shape[0]
    cin = w.shape[1]
    taps = w.numel() // (cout * cin)
    k = cin * taps
    inv_sigma = torch.empty(1, device=w.device, dtype=torch.float32)
    u_save = torch.empty_like(u)
    v_save = torch.empty_like(v)
    tmp = torch.empty(cout + k, device=w.device, dtype=torch.float32)
    call("dgmr_spectral_sigma", _p(w), _p(u), _p(v), _p(u_save), _p(v_save), _p(inv_sigma), _p(scratch), _p(tmp), cout, cin,
         taps, float(eps), int(bool(train)), _stream())
    return SNCall(inv_sigma, u_save.view(1, -1), v_save.view(1, -1), 1)


def weight_gram(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A = W W^T ([Cout, Cout]) of a conv / linear weight seen as the [Cout, K] matrix, on the MFMA conv kernel:
    the weight tensor is handed in both as the 'image' (Cout pixels of K channels) and as the 1x1 filter bank."""
    require_hip(w, "spectral-norm weight")
    cout = w.shape[0]
    k = w.numel() // cout
    a = out if out is not None else torch.empty(cout, cout, device=w.device, dtype=torch.float32)
    _launch_conv(w, _p(w), None, None, a, 1, 1, cout, 1, k, cout, 1, 1, 1)
    return a


def spectral_sigma_seq(w: torch.Tensor, gram: torch.Tensor, u: torch.Tensor, v: torch.Tensor, scratch: torch.Tensor, eps: float,
                       calls: int) -> SNCall:
    """`calls` consecutive train-mode calls of one module (one power iteration each) in one go; see dgmr_spectral_sigma_seq."""
    require_hip(w, "spectral-norm weight")
    cout, cin = w.shape[0], w.shape[1]
    taps = w.numel() // (cout * cin)
    k = cin * taps
    dev = w.device
    inv_sigma = torch.empty(calls, device=dev, dtype=torch.float32)
    u_hist = torch.empty(calls, cout, device=dev, dtype=torch.float32)
    v_hist = torch.empty(calls, k, device=dev, dtype=torch.float32)
    tmp = torch.empty(cout + calls, device=dev, dtype=torch.float32)
    call("dgmr_spectral_sigma_seq", _p(w), _p(gram), _p(u), _p(v), _p(u_hist), _p(v_hist), _p(inv_sigma), _p(scratch), _p(tmp),
         cout, cin, taps, float(eps), calls, _stream())
    return SNCall(inv_sigma, u_hist, v_hist, calls)


# ---------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------
@dataclass
class ConvSpec:
    """Everything about one conv call that is not a differentiable tensor."""

    upsample: bool = False  # nearest 2x on the input (common.py:142,148)
    pre_relu: bool = False  # relu on the input
    bn: Optional[BNState] = None  # BatchNorm+ReLU on the input
    sn: Optional[SNCall] = None  # spectral-norm call record (scale = 1/sigma)
    gamma_scale: bool = False  # scale tensor is a learnable scalar parameter (Attention.gamma)
    act_relu: bool = False  # relu on the output (F.relu(conv(..)), common.py:424)
    residual_up: bool = False  # the residual is at half resolution and is added with nearest-2x upsampling
    want_stats: bool = False  # also return per-tile partial sums (sum y, sum y^2) of the output: the next BatchNorm's batch statistics
    pool_out: bool = False  # y = AvgPool2d(2) / AvgPool3d(2) of the conv (+ residual at the pooled resolution): DBlock (common.py:233-237)

    @property
    def groups(self) -> int:
        """Number of module calls one launch covers (consecutive N/groups samples share a sigma)."""
        return self.sn.groups if self.sn is not None else 1


_flip_cache = {}


def _flipped_weight(w: torch.Tensor, coff: int = 0, cin: Optional[int] = None) -> torch.Tensor:
    """Weights of the transposed conv (optionally of the input-channel slice [coff, coff+cin)), cached until the parameter changes."""
    cout, cin_total = w.shape[0], w.shape[1]
    if cin is None:
        cin = cin_total
    key = (id(w), coff, cin)
    tag = _core.weight_tag(w)
    hit = _flip_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:  # the weakref guards against id() reuse after a parameter is freed
        return hit[1]
    ks = list(w.shape[2:])
    kd, kh, kw = ([1] + ks) if len(ks) == 2 else ks
    wt = torch.empty(cout * cin * kd * kh * kw, device=w.device, dtype=torch.float32)
    call("dgmr_conv_flip_weights", _p(w), _p(wt), cout, cin, kd, kh, kw, cin_total, coff, _stream())
    _flip_cache[key] = (tag, wt, weakref.ref(w, lambda _r, k=key: _flip_cache.pop(k, None)))  # freed with the parameter
    return wt


_split_cache = {}
_PRECISION_CODE = 0  # mirror of the library's mode (set_precision keeps it in sync): 0 means no pre-split weights are needed


def _split_planes(w: torch.Tensor, flipped: bool, coff: int = 0, cin: Optional[int] = None) -> Optional[torch.Tensor]:
    """bf16 (hi, lo) planes of a 3x3 (or 3x3x3) conv weight — optionally of its input-channel slice [coff, coff+cin) — or of the
    flipped / transposed version used by the data gradient; cached until the parameter changes.  None when the LDS-window kernels
    cannot take this conv (mode f32, not 3x3[x3], channel count % 8)."""
    ks = tuple(w.shape[2:])
    one = all(k == 1 for k in ks)  # 1x1 (1x1x1): the streaming 1x1 kernel (conv1x1.h) takes the same plane layout, one tap
    if _PRECISION_CODE == 0 or w.dim() not in (4, 5) or not (one or all(k == 3 for k in ks)):
        return None
    taps = 1 if one else (9 if w.dim() == 4 else 27)
    cout, cin_total = w.shape[0], w.shape[1]
    if cin is None:
        cin = cin_total
    rows_c, k_c = (cin, cout) if flipped else (cout, cin)  # rows of the matrix the kernel sees, and its input channels
    if k_c % 8:
        return None
    planes = _PLANES[_PRECISION_CODE]
    key = (id(w), flipped, coff, cin, planes)
    tag = _core.weight_tag(w)
    hit = _split_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    out = torch.empty(planes * cout * taps * cin, device=w.device, dtype=torch.int16)
    if flipped:  # the flipped slice is already dense
        call("dgmr_split_weights", _p(_flipped_weight(w, coff, cin)), _p(out), rows_c * taps, k_c, 0, 0, planes, 0, _stream())
    else:
        call("dgmr_split_weights", _p(w), _p(out), rows_c * taps, k_c, cin_total, coff, planes, 0, _stream())
    _split_cache[key] = (tag, out, weakref.ref(w, lambda _r, k=key: _split_cache.pop(k, None)))
    return out


_GRU_FUSE_GATES = __import__("os").environ.get("DGMR_GRU_FUSE", "1") != "0"  # measurement switch


def _split_planes_cat(ws: Sequence[torch.Tensor], coff: int, cin: int) -> Optional[torch.Tensor]:
    """bf16 planes of the input-channel slice [coff, coff+cin) of SEVERAL 3x3 conv weights stacked along the output-channel axis
    ([planes][sum Cout][9][cin]): the ConvGRU's read and update gate convs as ONE conv (DGMR_EPI_GRU_GATES2).  Cached until any of
    the parameters changes."""
    if _PRECISION_CODE == 0 or cin % 8 or any(w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) for w in ws):
        return None
    planes = _PLANES[_PRECISION_CODE]
    key = (tuple(id(w) for w in ws), "cat", coff, cin, planes)
    tag = tuple(_core.weight_tag(w) for w in ws)
    hit = _split_cache.get(key)
    if hit is not None and hit[0] == tag and all(r() is w for r, w in zip(hit[2], ws)):
        return hit[1]
    rows = sum(w.shape[0] for w in ws)
    stride = rows * 9 * cin
    out = torch.empty(planes * stride, device=ws[0].device, dtype=torch.int16)
    r0 = 0
    for w in ws:
        call("dgmr_split_weights", _p(w), out.data_ptr() + 2 * r0 * 9 * cin, w.shape[0] * 9, cin, w.shape[1], coff, planes, stride, _stream())
        r0 += w.shape[0]
    _split_cache[key] = (tag, out, tuple(weakref.ref(w, lambda _r, k=key: _split_cache.pop(k, None)) for w in ws))
    return out


_phase_cache = {}
_NO_PHASES = bool(int(__import__("os").environ.get("DGMR_NO_PHASES", "0")))  # measurement switch (tools/conv_bench.py)
_UP_WGRAD_SUMS = bool(int(__import__("os").environ.get("DGMR_UP_WGRAD_SUMS", "0")))  # A/B switch: pair-sum weight gradient of upsampling convs everywhere


def _phase_planes(w: torch.Tensor) -> Optional[torch.Tensor]:
    """bf16 (hi, lo) planes of the four parity ("phase") tap sums of a 3x3 conv that follows a nearest-2x upsample
    (dgmr_conv_args.w_phase); cached until the parameter changes.  None in exact-f32 mode (the upsampled conv then runs as written)."""
    if _PRECISION_CODE == 0 or _NO_PHASES or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[1] % 8:
        return None
    cout, cin = w.shape[0], w.shape[1]
    planes = _PLANES[_PRECISION_CODE]
    key = (id(w), "phase", planes)
    tag = _core.weight_tag(w)
    hit = _phase_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    sums = torch.empty(16 * cout * cin, device=w.device, dtype=torch.float32)
    call("dgmr_upsample_phase_weights", _p(w), _p(sums), cout, cin, _stream())
    out = torch.empty(planes * 16 * cout * cin, device=w.device, dtype=torch.int16)
    call("dgmr_split_weights", _p(sums), _p(out), 16 * cout, cin, 0, 0, planes, 0, _stream())
    _phase_cache[key] = (tag, out, weakref.ref(w, lambda _r, k=key: _phase_cache.pop(k, None)))
    return out


def _pool2_planes(w: torch.Tensor) -> Optional[torch.Tensor]:
    """The same for the data gradient of that conv: bf16 planes of the 4x4 stride-2 kernel "flipped 3x3 conv, then 2x2 sum pool",
    grouped by input-pixel parity (dgmr_conv_args.pool2 / w_phase)."""
    if _PRECISION_CODE == 0 or _NO_PHASES or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0] % 8:
        return None
    cout, cin = w.shape[0], w.shape[1]
    planes = _PLANES[_PRECISION_CODE]
    key = (id(w), "pool2", planes)
    tag = _core.weight_tag(w)
    hit = _phase_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    sums = torch.empty(16 * cout * cin, device=w.device, dtype=torch.float32)
    call("dgmr_pool2_phase_weights", _p(_flipped_weight(w)), _p(sums), cin, cout, _stream())
    out = torch.empty(planes * 16 * cout * cin, device=w.device, dtype=torch.int16)
    call("dgmr_split_weights", _p(sums), _p(out), 16 * cin, cout, 0, 0, planes, 0, _stream())
    _phase_cache[key] = (tag, out, weakref.ref(w, lambda _r, k=key: _phase_cache.pop(k, None)))
    return out


def _pool2_fwd_planes(w: torch.Tensor) -> Optional[torch.Tensor]:
    """bf16 planes of "3x3 conv, then 2x2 AVERAGE pool" as a 4x4 stride-2 kernel grouped by input-pixel parity (dgmr_conv_args.pool2 in
    the forward direction: a DBlock's last conv + pooling); 3x3x3 weights: 16 tap sums per depth tap.  The 1/4 is folded into the tap
    sums (a power of two: no rounding).  None where the window kernels cannot take the conv."""
    if _PRECISION_CODE == 0 or _NO_PHASES or w.dim() not in (4, 5) or any(k != 3 for k in w.shape[2:]) or w.shape[1] % 8:
        return None
    cout, cin = w.shape[0], w.shape[1]
    kd = 3 if w.dim() == 5 else 1
    planes = _PLANES[_PRECISION_CODE]
    key = (id(w), "pool2f", planes)
    tag = _core.weight_tag(w)
    hit = _phase_cache.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is w:
        return hit[1]
    sums = torch.empty(16 * kd * cout * cin, device=w.device, dtype=torch.float32)
    call("dgmr_pool2_phase_weights", _p(w), _p(sums), cout * kd, cin, _stream())  # rows (co, kd): [co][kd * 16 + t][ci]
    call("dgmr_axpby", _p(sums), None, _p(sums), 0.25, 0.0, sums.numel(), _stream())
    out = torch.empty(planes * sums.numel(), device=w.device, dtype=torch.int16)
    call("dgmr_split_weights", _p(sums), _p(out), 16 * kd * cout, cin, 0, 0, planes, 0, _stream())
    _phase_cache[key] = (tag, out, weakref.ref(w, lambda _r, k=key: _phase_cache.pop(k, None)))
    return out


def _kdims(w: torch.Tensor):
    ks = list(w.shape[2:])
    return tuple([1] + ks) if len(ks) == 2 else tuple(ks)


EPI_PLAIN, EPI_GRU_GATE, EPI_GRU_BLEND, EPI_GRU_GATES2 = 0, 1, 2, 3


def _launch_conv(x, w_ptr, bias, scale, y, n, d, h, w_, cin, cout, kd, kh, kw, *, upsample=False, pre_relu=False, pre_a=None,
                 pre_b=None, pre_group=1, residual=None, addend=None, mask_src=None, mask_a=None, mask_b=None, mask_group=1,
                 scale_group=None, act_relu=False, w_cin=0, w_coff=0, epi_mode=EPI_PLAIN, gru_h=None, gru_pu=None, pre_out=None,
                 device=None, w_split=None, residual_up=False, want_stats=False, w_phase=None, pool2=False, gates2=None):
    """`want_stats`: ask for the BatchNorm partial sums of the OUTPUT (dgmr_conv_args.stats_out); returns the [rows, 2, Cout] partials
    tensor, or None when the kernel the library dispatches for these arguments has no fused statistics.
    `pool2`: y is the 2x2 sum pool of the conv (dgmr_conv_args.pool2); returns NotImplemented - nothing launched - when the library has
    no single-pass kernel for these arguments.
    `gates2` = (scale2, bias2, addend2, y2, C) with epi_mode EPI_GRU_GATES2: the fused read + update gate launch; returns NotImplemented
    when the library cannot take it."""
    a = ConvArgs()
    if gates2 is not None:
        a.scale2, a.bias2, a.addend2, a.y2, a.gru_split = _p(gates2[0]), _p(gates2[1]), _p(gates2[2]), _p(gates2[3]), int(gates2[4])
    a.w_split = _p(w_split)
    a.w_phase = _p(w_phase)
    a.residual_up = int(bool(residual_up))
    a.w_cin, a.w_coff, a.epi_mode = w_cin, w_coff, epi_mode
    a.gru_h, a.gru_pu, a.pre_out = _p(gru_h), _p(gru_pu), _p(pre_out)
    ws = _splitk_ws(device if device is not None else (x.device if isinstance(x, torch.Tensor) else y.device))
    a.splitk_ws, a.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    a.x, a.w, a.bias, a.scale = _p(x), w_ptr, _p(bias), _p(scale)
    a.pre_a, a.pre_b, a.addend, a.residual = _p(pre_a), _p(pre_b), _p(addend), _p(residual)
    a.mask_src, a.mask_a, a.mask_b, a.y = _p(mask_src), _p(mask_a), _p(mask_b), _p(y)
    a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, d, h, w_, cin, cout
    a.KD, a.KH, a.KW = kd, kh, kw
    a.upsample, a.pre_relu = int(upsample), int(pre_relu)
    a.scale_group = scale_group if scale_group else n
    a.pre_group, a.mask_group = pre_group, mask_group
    a.act_relu = int(act_relu)
    a.pool2 = int(bool(pool2))
    if pool2:
        from ._lib import load

        if not load().dgmr_conv_pool2_supported(ctypes.byref(a)):
            return NotImplemented
    if gates2 is not None:
        from ._lib import load

        if not load().dgmr_conv_gates2_supported(ctypes.byref(a)):
            return NotImplemented
    partials = None
    if want_stats:
        from ._lib import load

        rows = int(load().dgmr_conv_stats_rows(ctypes.byref(a)))
        if rows > 0:
            partials = torch.empty(rows, 2, cout, device=y.device, dtype=torch.float32)
            a.stats_out = partials.data_ptr()
    call("dgmr_conv_fwd", ctypes.byref(a), _stream())
    return partials


class ConvFn(Function):
    """y = conv(pre(x), W) * scale + bias (+ residual), pre in {id, relu, BN+relu, nearest-2x∘those}."""

    @staticmethod
    def forward(ctx, x, w, bias, scale, residual, bn_gamma, bn_beta, spec: ConvSpec):
        require_hip(x)
        x = to_cl(x)
        n, cin, d, h, wd = _dims(x)
        if spec.upsample:
            h, wd = 2 * h, 2 * wd
        cout = w.shape[0]
        kd, kh, kw = _kdims(w)
        require_weight_layout(w)
        if w.shape[1] != cin:
            raise RuntimeError(f"conv: input has {cin} channels, weight expects {w.shape[1]}")
        oshape = (n, cout, h, wd) if x.dim() == 4 else (n, cout, d, h, wd)
        bn = spec.bn
        groups = spec.groups
        if n % groups:
            raise RuntimeError(f"conv: batch {n} is not divisible into {groups} spectral-norm call groups")
        if spec.pool_out:
            return ConvFn._forward_pooled(ctx, x, w, bias, scale, residual, spec, (n, cin, cout, d, h, wd, kd, kh, kw))
        y = empty_cl(oshape, x)
        if residual is not None:
            residual = to_cl(residual)
            want = (n, cout, h // 2, wd // 2) if spec.residual_up else (oshape)
            if tuple(residual.shape) != tuple(want):
                raise RuntimeError(f"conv: residual has shape {tuple(residual.shape)}, expected {tuple(want)}")
        partials = _launch_conv(x, _p(w), bias, scale, y, n, d, h, wd, cin, cout, kd, kh, kw, upsample=spec.upsample,
                                pre_relu=spec.pre_relu, pre_a=bn.a if bn else None, pre_b=bn.b if bn else None,
                                pre_group=bn.group_size if bn else 1, residual=residual, act_relu=spec.act_relu,
                                scale_group=n // groups, w_split=_split_planes(w, False), residual_up=spec.residual_up,
                                want_stats=spec.want_stats, w_phase=_phase_planes(w) if spec.upsample else None)
        ctx.spec = spec  # flags only are read from it in backward; its tensors are re-read from saved_tensors
        ctx.has_residual = residual is not None
        # parameters are kept as-is (checkpointing hands back DETACHED copies of saved tensors: .grad must land on the real ones)
        ctx.params = (w, bias, scale if spec.gamma_scale else None)
        sn = spec.sn
        # Every tensor the backward reads goes through save_for_backward: under activation checkpointing
        # (dgmr/dgmr.py:150,176) the saved tensors are replaced by those of the RECOMPUTED forward, whose sigma / u / v
        # and BatchNorm statistics differ from the first forward's; a tensor stashed on ctx would not be.
        ctx.save_for_backward(x, scale, y if spec.act_relu else None, sn.u if sn else None, sn.v if sn else None,
                              bn.a if bn else None, bn.b if bn else None, bn.mean if bn else None, bn.rstd if bn else None)
        ctx.geom = (n, cin, cout, d, h, wd, kd, kh, kw)
        if spec.want_stats:
            if partials is not None:
                ctx.mark_non_differentiable(partials)
            return y, partials
        return y

    @staticmethod
    def _forward_pooled(ctx, x, w, bias, scale, residual, spec: ConvSpec, geom):
        """conv + AvgPool (+ residual at the pooled resolution), DBlock's tail (common.py:233-237).  In the bf16 modes the window kernel
        evaluates "3x3 conv, then 2x2 average" as ONE 4x4 stride-2 pass over the input's pixel-parity planes (16 instead of 36 multiply
        steps per input pixel; the full-resolution conv output is never written or read back); a 3x3x3 conv gets the spatial half of its
        AvgPool3d that way and the depth pair average as a streaming pass over the small map.  Elsewhere: the conv, then the pooling
        kernel.  The backward pass is the composite's: dy is spread back over the windows first, the rest is the plain conv's."""
        n, cin, cout, d, h, wd, kd, kh, kw = geom
        if spec.upsample or spec.act_relu or spec.want_stats or spec.bn is not None or spec.residual_up:
            raise RuntimeError("conv: pool_out combines with pre_relu / a pooled-resolution residual only")
        is3d = x.dim() == 5
        pd = 2 if is3d else 1
        if h < 2 or wd < 2 or (is3d and d < 2):
            raise RuntimeError(f"conv: pool_out needs maps of at least one pooling window (got {d}x{h}x{wd})")
        # odd maps floor like nn.AvgPool2d / AvgPool3d (the last row / column / plane is dropped: 6 -> 3 -> 1 for 96x96 inputs); only the
        # fused 4x4 stride-2 pass needs whole spatial windows - odd maps take the conv + pooling-kernel branch below (dgmr_pool_bwd zero-fills)
        even = h % 2 == 0 and wd % 2 == 0  # (an odd DEPTH stays on the fused path: dgmr_pool_depth2 drops the last plane)
        oshape = (n, cout, d // pd, h // 2, wd // 2) if is3d else (n, cout, h // 2, wd // 2)
        if residual is not None:
            residual = to_cl(residual)
            if tuple(residual.shape) != oshape:
                raise RuntimeError(f"conv: residual has shape {tuple(residual.shape)}, expected {oshape}")
        groups = spec.groups
        y = empty_cl(oshape, x)
        done = NotImplemented
        w_pool = _pool2_fwd_planes(w) if (even and (kd, kh, kw) in ((1, 3, 3), (3, 3, 3))) else None
        if w_pool is not None:
            sp = empty_cl((n, cout, d, h // 2, wd // 2), x) if is3d else y
            done = _launch_conv(x, _p(w), bias, scale, sp, n, d, h, wd, cin, cout, kd, kh, kw, pre_relu=spec.pre_relu,
                                residual=None if is3d else residual, scale_group=n // groups, w_split=_split_planes(w, False),
                                w_phase=w_pool, pool2=True)
            if done is not NotImplemented and is3d:
                call("dgmr_pool_depth2", _p(sp), _p(residual), _p(y), n, d, (h // 2) * (wd // 2) * cout, _stream())
        if done is NotImplemented:
            full = empty_cl((n, cout, d, h, wd) if is3d else (n, cout, h, wd), x)
            _launch_conv(x, _p(w), bias, scale, full, n, d, h, wd, cin, cout, kd, kh, kw, pre_relu=spec.pre_relu,
                         scale_group=n // groups, w_split=_split_planes(w, False))
            call("dgmr_pool_fwd", _p(full), _p(residual), _p(y), n, d, h, wd, cout, pd, 0.0, None, None, None, 1, _stream())
        ctx.spec = spec
        ctx.has_residual = residual is not None
        ctx.params = (w, bias, scale if spec.gamma_scale else None)
        sn = spec.sn
        ctx.save_for_backward(x, scale, None, sn.u if sn else None, sn.v if sn else None, None, None, None, None)
        ctx.geom = geom
        return y

    @staticmethod
    def backward(ctx, dy, _dpartials=None):
        spec: ConvSpec = ctx.spec
        x, scale, y_act, sn_u, sn_v, bn_a, bn_b, bn_mean, bn_rstd = ctx.saved_tensors
        w, bias, scale_param = ctx.params
        n, cin, cout, d, h, wd, kd, kh, kw = ctx.geom
        dy = to_cl(dy)
        dev = dy.device
        dy_pooled = None
        if spec.pool_out:  # gradient of the pooling first (dy / window over every window): the rest is the plain conv's backward
            dy_pooled = dy
            dy = empty_cl((n, cout, h, wd) if x.dim() == 4 else (n, cout, d, h, wd), dy_pooled)
            call("dgmr_pool_bwd", _p(dy_pooled), _p(dy), n, d, h, wd, cout, 2 if x.dim() == 5 else 1, 0.0, _stream())
        if y_act is not None:  # relu epilogue (never combined with a residual on this path)
            assert not ctx.has_residual
            dz = torch.empty_like(dy)
            call("dgmr_relu_bwd", _p(dy), _p(y_act), _p(dz), dy.numel(), _stream())
            dy = dz
        m = n * d * h * wd
        k = kd * kh * kw * cin
        bn = spec.bn
        st = _stream()
        # ---- bias: column sums of dy ride along in the weight-gradient kernel; standalone only when W is frozen ----
        want_bias = bias is not None and bias.requires_grad
        if want_bias and not w.requires_grad:
            tmp = colsum_tmp(m, cout, dev)
            call("dgmr_colsum", _p(dy), _p(grad_buffer(bias)), _p(tmp), m, cout, 1, st)
        # ---- weight (and scale) ----
        groups = spec.groups
        taps = kd * kh * kw
        if w.requires_grad:
            def weight_grad():
                st = _stream()  # (the side stream when run there)
                wa = WgradArgs()
                wa.pre_a, wa.pre_b = (_p(bn_a), _p(bn_b)) if bn else (None, None)
                wa.pre_relu, wa.pre_group = int(spec.pre_relu), (bn.group_size if bn else 1)
                wa.groups = groups
                # (where the wave-specialised LDS-window weight gradient applies - whole rows of 32 output pixels, or 16-pixel-wide maps -
                #  it takes the upsampling conv as it is, nearest-2x fused into its loads, at 330 - 350 TF: 2.3 ms for up_g4.first
                #  against ~20 ms for the sums kernel, which writes a 9-plane map of 4 GB, plus the GEMM over it)
                window_wgrad = (wd % 32 == 0 and h % 2 == 0) or (wd == 16 and h % 4 == 0)
                if (spec.upsample and _PRECISION_CODE != 0 and not _NO_PHASES and (kd, kh, kw) == (1, 3, 3) and d == 1
                        and (_UP_WGRAD_SUMS or not window_wgrad)):
                    # upsampling conv, bf16 modes: sum the 2x2 pixels of dy that meet each INPUT pixel under each tap (9 planes), then the
                    # gradient is a 1x1 problem on the low-resolution map - a quarter of the multiply steps (dgmr_upsample_wgrad_sums)
                    z9 = _scratch(n * (h // 2) * (wd // 2) * 9 * cout, dev, "z9")
                    call("dgmr_upsample_wgrad_sums", _p(dy), _p(z9), n, h // 2, wd // 2, cout, st)
                    wa.x, wa.dy = _p(x), _p(z9)
                    wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = n, 1, h // 2, wd // 2, cin, 9 * cout
                    wa.KD, wa.KH, wa.KW, wa.upsample = 1, 1, 1, 0
                    # bias: the centre-tap plane (ky = kx = 1) holds every pixel of dy exactly once - its column sums ride in the kernel
                    z9_bias = torch.zeros(9 * cout, device=dev, dtype=torch.float32) if want_bias else None
                    wa.bias_grad = _p(z9_bias)
                else:
                    z9_bias = None
                    wa.x, wa.dy = _p(x), _p(dy)
                    wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = n, d, h, wd, cin, cout
                    wa.KD, wa.KH, wa.KW, wa.upsample = kd, kh, kw, int(spec.upsample)
                    wa.bias_grad = _p(grad_buffer(bias)) if want_bias else None
                call("dgmr_conv_wgrad_plan", ctypes.byref(wa))  # slab count: depends on which kernel the library will pick
                ns = wa.nsplit
                partial = torch.empty(ns * cout * k, device=dev, dtype=torch.float32)
                wa.partial = _p(partial)
                rows_ws = bias_rows(wa, dev)  # deterministic mode: the slabs' bias sums in rows of their own (kept alive past the launch)
                call("dgmr_conv_wgrad", ctypes.byref(wa), st)
                del rows_ws
                if z9_bias is not None:
                    grad_buffer(bias).add_(z9_bias.view(cout, 9)[:, 4])
                gw = grad_buffer(w)
                g = torch.empty(cout * k, device=dev, dtype=torch.float32)
                if scale is None:
                    call("dgmr_wgrad_reduce", _p(partial), ns, 1, cout * k, None, None, _p(g), None, st)
                    call("dgmr_sn_wgrad_finalize", _p(g), _p(gw), None, None, None, None, cout, cin, taps, 1, 1, st)
                else:
                    # g = sum_q P_q / sigma_q ; dot[q] = <P_q, W>   (P_q: raw weight gradient over the rows of call q)
                    dot = dot_buffer(groups, dev)
                    call("dgmr_wgrad_reduce", _p(partial), ns, groups, cout * k, _p(w), _p(scale), _p(g), _p(dot), st)
                    if spec.sn is not None:
                        call("dgmr_sn_wgrad_finalize", _p(g), _p(gw), _p(dot), _p(scale), _p(sn_u), _p(sn_v), cout, cin, taps, groups, 1, st)
                    else:  # learnable scalar gain: d scale = <P, W>, dW = scale * P
                        if scale_param is not None and scale_param.requires_grad:
                            gs = grad_buffer(scale_param)
                            call("dgmr_axpby", _p(gs), _p(dot), _p(gs), 1.0, 1.0, 1, st)
                        call("dgmr_sn_wgrad_finalize", _p(g), _p(gw), None, None, None, None, cout, cin, taps, 1, 1, st)

            if _WGRAD_STREAM and ctx.needs_input_grad[0]:
                _on_side_stream(dev, weight_grad, (x, dy, scale, bn_a, bn_b, sn_u, sn_v), lane=0 if spec.upsample else None,
                                cost=float(m) * k * cout * (4.0 / 9.0 if spec.upsample else 1.0))
            else:
                weight_grad()
        # ---- input ----
        dx = None
        if ctx.needs_input_grad[0]:
            wt = _flipped_weight(w)
            g_sums = NotImplemented
            if spec.upsample:
                # flipped conv at full resolution, then the 2x2 window sum (nearest-2x backwards): one pass over the parity planes of dy
                w_pool = _pool2_planes(w)
                if w_pool is not None:
                    g = empty_cl(x.shape, dy)
                    g_sums = _launch_conv(dy, _p(wt), None, scale, g, n, d, h, wd, cout, cin, kd, kh, kw,
                                          mask_src=x if (bn or spec.pre_relu) else None, mask_a=bn_a if bn else None,
                                          mask_b=bn_b if bn else None, mask_group=bn.group_size if bn else 1, scale_group=n // groups,
                                          w_split=_split_planes(w, True), w_phase=w_pool, pool2=True, want_stats=bn is not None)
            if g_sums is not NotImplemented:
                pass
            elif spec.upsample:
                g_sums = None
                hi = empty_cl((n, cin, h, wd) if x.dim() == 4 else (n, cin, d, h, wd), dy)
                _launch_conv(dy, _p(wt), None, scale, hi, n, d, h, wd, cout, cin, kd, kh, kw, scale_group=n // groups,
                             w_split=_split_planes(w, True))
                g = empty_cl(x.shape, dy)
                call("dgmr_pool_fwd", _p(hi), None, _p(g), n, d, h, wd, cin, 1, 1.0, _p(x) if (bn or spec.pre_relu) else None,
                     _p(bn_a) if bn else None, _p(bn_b) if bn else None, bn.group_size if bn else 1, st)
            else:
                g = empty_cl(x.shape, dy)
                # behind a BatchNorm the epilogue also leaves per-tile (sum g, sum g * x): BatchNorm's backward reduction
                g_sums = _launch_conv(dy, _p(wt), None, scale, g, n, d, h, wd, cout, cin, kd, kh, kw,
                                      mask_src=x if (bn or spec.pre_relu) else None, mask_a=bn_a if bn else None,
                                      mask_b=bn_b if bn else None, mask_group=bn.group_size if bn else 1, scale_group=n // groups,
                                      w_split=_split_planes(w, True), want_stats=bn is not None)
            if bn is None:
                dx = g
            else:
                c = cin
                r = x.numel() // (c * bn.groups)
                from_partials = g_sums is not None and g_sums.shape[0] % bn.groups == 0
                sums = sums_buffer(bn.groups, r, c, dev, row_blocks=not from_partials)
                if from_partials:
                    call("dgmr_bn_partial_reduce", _p(g_sums), _p(sums), bn.groups, g_sums.shape[0] // bn.groups, c, st)
                    call("dgmr_bn_bwd_center", _p(sums), _p(bn_mean), _p(bn_rstd), bn.groups, c, st)
                else:
                    call("dgmr_bn_bwd_reduce", _p(g), _p(x), _p(bn_mean), _p(bn_rstd), _p(sums), bn.groups, r, c, st)
                dx = empty_cl(x.shape, dy)
                dgam = grad_buffer(bn.gamma) if (bn.gamma is not None and bn.gamma.requires_grad) else None
                dbet = grad_buffer(bn.beta) if (bn.beta is not None and bn.beta.requires_grad) else None
                call("dgmr_bn_bwd_apply", _p(g), _p(x), _p(bn_mean), _p(bn_rstd), _p(bn.gamma), _p(sums), None, _p(dx), _p(dgam),
                     _p(dbet), bn.groups, r, c, int(bn.train), st)
        d_res = None
        if ctx.has_residual and dy_pooled is not None:
            d_res = dy_pooled
        elif ctx.has_residual:
            if spec.residual_up:  # backward of the nearest-2x upsample: sum over the 2x2 window
                d_res = empty_cl((n, cout, h // 2, wd // 2), dy)
                call("dgmr_pool_fwd", _p(dy), None, _p(d_res), n, 1, h, wd, cout, 1, 1.0, None, None, None, 1, st)
            else:
                d_res = dy
        return dx, None, None, None, d_res, None, None, None


def call_nsplit(m: int, cout: int, k: int, groups: int = 1) -> int:
    from ._lib import load

    return int(load().dgmr_conv_wgrad_nsplit(m, cout, k, groups))


class HeadFn(Function):
    """The sampler's output layer, relu(BatchNorm(x)) -> SN 1x1 conv to 4 channels (generators.py:159-166), as three streaming fp32
    kernels (dgmr_head_*): 28 M pixels x 48 channels at the paper size, HBM-bound.  The data gradient of the conv is never written:
    pass 1 takes BatchNorm's backward sums, the raw weight gradient and the bias gradient from it in registers, pass 2 recomputes it
    and applies BatchNorm's backward.  Same contract as ConvFn (parameter gradients accumulate straight into .grad)."""

    @staticmethod
    def forward(ctx, x, w, bias, scale, bn_gamma, bn_beta, spec: ConvSpec):
        require_hip(x)
        x = to_cl(x)
        require_weight_layout(w)
        n, c, _, h, wd = _dims(x)
        bn = spec.bn
        m, ppg = n * h * wd, bn.group_size * h * wd
        y = empty_cl((n, 4, h, wd), x)
        call("dgmr_head_fwd", _p(x), _p(bn.a), _p(bn.b), _p(w), _p(bias), _p(scale), _p(y), m, ppg, c, _stream())
        sn = spec.sn
        ctx.spec = spec
        ctx.params = (w, bias)
        ctx.save_for_backward(x, scale, sn.u if sn else None, sn.v if sn else None, bn.a, bn.b, bn.mean, bn.rstd)
        ctx.geom = (n, c, h, wd, m, ppg)
        return y

    @staticmethod
    def backward(ctx, dy):
        from ._lib import load

        spec: ConvSpec = ctx.spec
        x, scale, sn_u, sn_v, bn_a, bn_b, bn_mean, bn_rstd = ctx.saved_tensors
        w, bias = ctx.params
        n, c, h, wd, m, ppg = ctx.geom
        bn = spec.bn
        dy = to_cl(dy)
        dev, st = dy.device, _stream()
        groups = m // ppg
        nblk = int(load().dgmr_head_blocks(m, ppg, c))
        bn_part = torch.empty(nblk, 2, c, device=dev, dtype=torch.float32)
        w_part = torch.empty(nblk, 4, c, device=dev, dtype=torch.float32)
        b_part = torch.empty(nblk, 4, device=dev, dtype=torch.float32)
        call("dgmr_head_bwd_sums", _p(x), _p(bn_a), _p(bn_b), _p(w), _p(scale), _p(dy), _p(bn_part), _p(w_part), _p(b_part), m, ppg, c, st)
        if bias is not None and bias.requires_grad:
            grad_buffer(bias).add_(b_part.sum(0))
        if w.requires_grad:
            gw = grad_buffer(w)
            g = torch.empty(4 * c, device=dev, dtype=torch.float32)
            if scale is None:
                call("dgmr_wgrad_reduce", _p(w_part), nblk, 1, 4 * c, None, None, _p(g), None, st)
                call("dgmr_sn_wgrad_finalize", _p(g), _p(gw), None, None, None, None, 4, c, 1, 1, 1, st)
            else:
                dot = dot_buffer(groups, dev)
                call("dgmr_wgrad_reduce", _p(w_part), nblk, groups, 4 * c, _p(w), _p(scale), _p(g), _p(dot), st)
                call("dgmr_sn_wgrad_finalize", _p(g), _p(gw), _p(dot), _p(scale), _p(sn_u), _p(sn_v), 4, c, 1, groups, 1, st)
        sums = sums_buffer(groups, nblk // groups, c, dev, row_blocks=False)
        call("dgmr_bn_partial_reduce", _p(bn_part), _p(sums), groups, nblk // groups, c, st)
        call("dgmr_bn_bwd_center", _p(sums), _p(bn_mean), _p(bn_rstd), groups, c, st)
        dx = empty_cl(x.shape, dy)
        dgam = grad_buffer(bn.gamma) if (bn.gamma is not None and bn.gamma.requires_grad) else None
        dbet = grad_buffer(bn.beta) if (bn.beta is not None and bn.beta.requires_grad) else None
        call("dgmr_head_bwd_apply", _p(x), _p(bn_a), _p(bn_b), _p(w), _p(scale), _p(dy), _p(bn_mean), _p(bn_rstd), _p(bn.gamma), _p(sums),
             _p(dx), _p(dgam), _p(dbet), m, ppg, c, int(bn.train), st)
        return dx, None, None, None, None, None, None


def _head_applies(x, w, scale, residual, spec: ConvSpec) -> bool:
    bn = spec.bn
    if bn is None or x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0] != 4 or residual is not None:
        return False
    if spec.upsample or spec.act_relu or spec.want_stats or spec.gamma_scale or w.shape[1] != x.shape[1]:
        return False
    if (spec.sn is not None and spec.sn.groups != bn.groups) or (spec.sn is None and scale is not None):
        return False
    from ._lib import load

    n, c, h, wd = x.shape
    return int(load().dgmr_head_blocks(n * h * wd, bn.group_size * h * wd, c)) > 0


def conv(x, w, bias=None, scale=None, residual=None, spec: Optional[ConvSpec] = None):
    spec = spec or ConvSpec()
    bn = spec.bn
    if _head_applies(x, w, scale, residual, spec):
        return HeadFn.apply(x, w, bias, scale, bn.gamma, bn.beta, spec)
    return ConvFn.apply(x, w, bias, scale, residual, bn.gamma if bn else None, bn.beta if bn else None, spec)


# ---------------------------------------------------------------------------------------------------
# ConvGRU gating (dgmr/layers/ConvGRU.py:69-85)
# ---------------------------------------------------------------------------------------------------
class GruGateFn(Function):
    @staticmethod
    def forward(ctx, pr, h):
        pr, h = to_cl(pr), to_cl(h)
        require_hip(pr)
        out = torch.empty_like(pr)
        call("dgmr_gru_gate_fwd", _p(pr), _p(h), _p(out), pr.numel(), _stream())
        ctx.save_for_backward(pr, h)
        return out

    @staticmethod
    def backward(ctx, d):
        pr, h = ctx.saved_tensors
        d = to_cl(d)
        dpr, dh = torch.empty_like(pr), torch.empty_like(pr)
        call("dgmr_gru_gate_bwd", _p(d), _p(pr), _p(h), _p(dpr), _p(dh), pr.numel(), _stream())
        return dpr, dh


class GruBlendFn(Function):
    @staticmethod
    def forward(ctx, pu, h, pc):
        pu, h, pc = to_cl(pu), to_cl(h), to_cl(pc)
        require_hip(pu)
        out = torch.empty_like(pu)
        call("dgmr_gru_blend_fwd", _p(pu), _p(h), _p(pc), _p(out), pu.numel(), _stream())
        ctx.save_for_backward(pu, h, pc)
        return out

    @staticmethod
    def backward(ctx, d):
        pu, h, pc = ctx.saved_tensors
        d = to_cl(d)
        dpu, dh, dpc = torch.empty_like(pu), torch.empty_like(pu), torch.empty_like(pu)
        call("dgmr_gru_blend_bwd", _p(d), _p(pu), _p(h), _p(pc), _p(dpu), _p(dh), _p(dpc), pu.numel(), _stream())
        return dpu, dh, dpc


gru_gate = GruGateFn.apply
gru_blend = GruBlendFn.apply


class ConvGRUFn(Function):
    """A whole ConvGRU layer over T steps (dgmr/layers/ConvGRU.py:57-85,102-111) with hand-written backward-through-time.

    The reference convolves torch.cat([x_t, h]) three times per step.  Convolution is linear in its input channels, so each
    conv splits into an x part and an h part.  The x parts of all T steps do not depend on the recurrence: they are three
    batched launches up front (M = T*B*h*w rows).  Only the h parts (1/3 of K) stay on the sequential path, as three launches
    per step whose epilogues carry the gating:  r*h = sigmoid(pre_r)*h  and  h' = u*h + (1-u)*relu(pre_c)  — no torch.cat, no
    separate gate kernels.  Small-M steps are split over K inside the library to fill the chip.  The backward sweeps t = T-1..0
    with three data-gradient convs per step; the x-part data gradients and all weight gradients (with the per-step
    spectral-norm chain rule) are batched over T afterwards.

    x_shared: x_all is ONE sample [1, Cx, h, w] that every sample at every step receives (the sampler's first ConvGRU is fed
    `[repeat(latent, B)] * T`, generators.py:146-149).  Its x parts are then three convs of a single 8x8 map instead of T*B
    identical ones; the backward sums the gate gradients over the samples first (convolution is linear), so the x-part data
    and weight gradients run on T maps instead of T*B.

    draws: the batch holds `draws` generator draws (forward calls of the reference), draw-major inside every step: step tensors are
    [draws * B', ...], the spectral-norm records carry steps * draws call groups in [step][draw] order (CallLayout time-major), and
    a shared x is one map PER DRAW ([draws, Cx, h, w]: every draw has its own latent).
    """

    @staticmethod
    def forward(ctx, x_all, h0, params, seqs, steps: int, x_shared: bool = False, draws: int = 1, anchor=None):
        """`anchor`: one of the layer's parameters, passed as a tensor argument only so that the output joins the autograd graph when
        neither x nor h0 requires a gradient (a stand-alone layer fed with data, tests/test_model.py:67-82); parameter gradients
        themselves are accumulated by the kernels into `param.grad`."""
        require_hip(x_all)
        require_hip(h0, "initial state")
        x_all, h0 = to_cl(x_all), to_cl(h0)
        wr, br, wu, bu, wc, bc = params
        for w_ in (wr, wu, wc):
            require_weight_layout(w_, "ConvGRU weight")
        T = steps
        nx, cx, hh, ww = x_all.shape
        b, ch = h0.shape[0], h0.shape[1]
        tb = T * b
        if nx != (draws if x_shared else tb) or b % draws or wr.shape[1] != cx + ch or wr.shape[0] != ch:
            raise RuntimeError(f"ConvGRU: x {tuple(x_all.shape)} / h0 {tuple(h0.shape)} / weight {tuple(wr.shape)} do not fit T={T}, "
                               f"draws={draws}{' (shared x)' if x_shared else ''}")
        bs = b // draws  # samples per draw = samples per spectral-norm call group within a step
        kh, kw = wr.shape[2], wr.shape[3]
        dev = x_all.device
        n_step = b * ch * hh * ww  # floats per step tensor
        ct = cx + ch

        def step_ptr(t_: torch.Tensor, t: int) -> int:
            return t_.data_ptr() + 4 * n_step * t

        def scale_ptr(sn: SNCall, t: int) -> int:  # the `draws` sigmas of step t are consecutive (groups in [step][draw] order)
            return sn.inv_sigma.data_ptr() + (4 * t * draws if sn.groups > 1 else 0)

        def sgroup(sn: SNCall) -> int:
            return bs if sn.groups > 1 else b

        # x parts of the three convs for every step: raw sums (scale and bias are applied with the h part)
        xparts = []
        for w in (wr, wu, wc):
            xp = empty_cl((nx, ch, hh, ww), x_all)
            _launch_conv(x_all, _p(w), None, None, xp, nx, 1, hh, ww, cx, ch, 1, kh, kw, w_cin=ct, w_coff=0,
                         w_split=_split_planes(w, False, 0, cx))
            if x_shared:  # one map per draw: a copy for each of its samples, read by every step
                x1, xp = xp, empty_cl((b, ch, hh, ww), x_all)
                call("dgmr_repeat_interleave", _p(x1), _p(xp), draws, n_step // b, bs, _stream())
            xparts.append(xp)
        xr, xu, xc = xparts
        if x_shared:
            def x_ptr(t_: torch.Tensor, t: int) -> int:
                return t_.data_ptr()
        else:
            x_ptr = step_ptr
        buf = empty_cl(((T + 1) * b, ch, hh, ww), x_all)  # h_{-1} = h0, h_0, ..., h_{T-1}
        _copy(_p(h0), _p(buf), n_step)
        # forwards without a graph (the discriminator passes' generator forward, the first pass of the checkpointed draws, eval): the
        # gate pre-activations and r*h of earlier steps are only read by the backward - not stored, one step of scratch instead
        keep = any(ctx.needs_input_grad) or _GRU_KEEP_ALWAYS
        if keep:
            pr, pu, pc, rh = (empty_cl((tb, ch, hh, ww), x_all) for _ in range(4))
        else:
            pu, rh = (empty_cl((b, ch, hh, ww), x_all) for _ in range(2))
            pr = pc = None

        def kept(t_: Optional[torch.Tensor], t: int):
            return None if t_ is None else (step_ptr(t_, t) if keep else t_.data_ptr())

        sr, su, sc = seqs
        spr, spu, spc = (_split_planes(w, False, cx, ch) for w in (wr, wu, wc))  # h halves as bf16 planes (bf16 modes, big maps)
        # read and update gate convolve the same h (ConvGRU.py:69-76): one launch with 2 ch output columns whenever the LDS-DMA window
        # kernel takes the step (DGMR_EPI_GRU_GATES2: the halo of h is staged once, a third fewer launches on the recurrent path)
        sp_ru = _split_planes_cat((wr, wu), cx, ch) if (_GRU_FUSE_GATES and (kh, kw) == (3, 3) and sgroup(sr) == sgroup(su)) else None
        fused = sp_ru is not None
        for t in range(T):
            hp, out = step_ptr(buf, t), step_ptr(buf, t + 1)
            if fused:
                r_ = _launch_conv(hp, _p(wr), br, scale_ptr(sr, t), kept(rh, t), b, 1, hh, ww, ch, 2 * ch, 1, kh, kw, addend=x_ptr(xr, t),
                                  epi_mode=EPI_GRU_GATES2, gru_h=hp, pre_out=kept(pr, t), device=dev, w_split=sp_ru, scale_group=sgroup(sr),
                                  gates2=(scale_ptr(su, t), bu, x_ptr(xu, t), kept(pu, t), ch))
                fused = r_ is not NotImplemented  # (decided by the first step: the geometry is the same for all of them)
            if not fused:
                _launch_conv(hp, _p(wr), br, scale_ptr(sr, t), kept(rh, t), b, 1, hh, ww, ch, ch, 1, kh, kw, w_cin=ct, w_coff=cx,
                             addend=x_ptr(xr, t), epi_mode=EPI_GRU_GATE, gru_h=hp, pre_out=kept(pr, t), device=dev, w_split=spr,
                             scale_group=sgroup(sr))
                _launch_conv(hp, _p(wu), bu, scale_ptr(su, t), kept(pu, t), b, 1, hh, ww, ch, ch, 1, kh, kw, w_cin=ct, w_coff=cx,
                             addend=x_ptr(xu, t), device=dev, w_split=spu, scale_group=sgroup(su))
            _launch_conv(kept(rh, t), _p(wc), bc, scale_ptr(sc, t), out, b, 1, hh, ww, ch, ch, 1, kh, kw, w_cin=ct, w_coff=cx,
                         addend=x_ptr(xc, t), epi_mode=EPI_GRU_BLEND, gru_h=hp, gru_pu=kept(pu, t), pre_out=kept(pc, t),
                         device=dev, w_split=spc, scale_group=sgroup(sc))
        ctx.params = params
        ctx.geom = (T, b, cx, ch, hh, ww, kh, kw, draws)
        ctx.x_shared = x_shared
        ctx.groups = tuple(q.groups for q in seqs)
        if keep:
            ctx.save_for_backward(x_all, buf, pr, pu, pc, rh, sr.inv_sigma, sr.u, sr.v, su.inv_sigma, su.u, su.v, sc.inv_sigma, sc.u, sc.v)
        return buf[b:]

    @staticmethod
    def backward(ctx, dout_all):
        (x_all, buf, pr, pu, pc, rh, isr, ur, vr, isu, uu, vu, isc, uc, vc) = ctx.saved_tensors
        wr, br, wu, bu, wc, bc = ctx.params
        T, b, cx, ch, hh, ww, kh, kw, draws = ctx.geom
        bs = b // draws
        gr, gu, gc = ctx.groups
        dout_all = to_cl(dout_all)
        dev = dout_all.device
        flush_deferred()  # (DGMR_WGRAD_DEFER: the held-back weight gradients of this level's G-blocks start beside the recurrent chain)
        st = _stream()
        tb = T * b
        n_step = b * ch * hh * ww
        ct = cx + ch
        taps = kh * kw

        def step_ptr(t_: torch.Tensor, t: int) -> int:
            return t_.data_ptr() + 4 * n_step * t

        def scale_ptr(inv_sigma: torch.Tensor, groups: int, t: int) -> int:
            return inv_sigma.data_ptr() + (4 * t * draws if groups > 1 else 0)

        def sgroup(groups: int) -> int:
            return bs if groups > 1 else b

        dpr, dpu, dpc = (empty_cl((tb, ch, hh, ww), dout_all) for _ in range(3))
        # scratch of one step each
        d_tot, dh_a, dh_b, d_rh, c1, c2 = (torch.empty(n_step, device=dev, dtype=torch.float32) for _ in range(6))
        dh_next = torch.empty(n_step, device=dev, dtype=torch.float32)
        wt_rh, wt_uh, wt_ch = (_flipped_weight(w, cx, ch) for w in (wr, wu, wc))
        sp_rh, sp_uh, sp_ch = (_split_planes(w, True, cx, ch) for w in (wr, wu, wc))
        have_next = False
        for t in reversed(range(T)):
            hp = step_ptr(buf, t)
            if have_next:
                call("dgmr_axpby", step_ptr(dout_all, t), _p(dh_next), _p(d_tot), 1.0, 1.0, n_step, st)
                d = _p(d_tot)
            else:
                d = step_ptr(dout_all, t)
            call("dgmr_gru_blend_bwd", d, step_ptr(pu, t), hp, step_ptr(pc, t), step_ptr(dpu, t), _p(dh_a), step_ptr(dpc, t), n_step, st)
            # through the candidate conv to r*h, then through the read gate
            _launch_conv(step_ptr(dpc, t), _p(wt_ch), None, scale_ptr(isc, gc, t), d_rh, b, 1, hh, ww, ch, ch, 1, kh, kw, device=dev,
                         w_split=sp_ch, scale_group=sgroup(gc))
            call("dgmr_gru_gate_bwd", _p(d_rh), step_ptr(pr, t), hp, step_ptr(dpr, t), _p(dh_b), n_step, st)
            # dh = dh_a + dh_b + convT(dpr / sigma_r, W_rh) + convT(dpu / sigma_u, W_uh)
            _launch_conv(step_ptr(dpr, t), _p(wt_rh), None, scale_ptr(isr, gr, t), c1, b, 1, hh, ww, ch, ch, 1, kh, kw, residual=dh_a,
                         device=dev, w_split=sp_rh, scale_group=sgroup(gr))
            _launch_conv(step_ptr(dpu, t), _p(wt_uh), None, scale_ptr(isu, gu, t), c2, b, 1, hh, ww, ch, ch, 1, kh, kw, residual=c1,
                         device=dev, w_split=sp_uh, scale_group=sgroup(gu))
            call("dgmr_axpby", _p(c2), _p(dh_b), _p(dh_next), 1.0, 1.0, n_step, st)
            have_next = True
        dh0 = None
        if ctx.needs_input_grad[1]:
            dh0 = empty_cl((b, ch, hh, ww), dout_all)
            _copy(_p(dh_next), _p(dh0), n_step)
        x_shared = ctx.x_shared
        n_img = ch * hh * ww
        TD = T * draws
        if x_shared:
            # the samples of a (step, draw) share x: sum their gate gradients first -> [T * draws, ch, h, w]
            dsum = []
            for dp in (dpr, dpu, dpc):
                ds = empty_cl((TD, ch, hh, ww), dout_all)
                call("dgmr_group_rowsum", _p(dp), None, _p(ds), TD, bs, n_img, 1, 1, st)
                dsum.append(ds)
            x_rep = empty_cl((TD, cx, hh, ww), dout_all)  # the shared maps once per step, as the weight gradient's "input"
            call("dgmr_repeat_rows", _p(x_all), _p(x_rep), draws * cx * hh * ww, T, st)
        # ---- weight / bias gradients, batched over T: off the critical path, beside the x-part data gradients and the layers below ----
        def weight_grads():
            st = _stream()  # (the side stream when run there)
            hprev_all = buf  # rows [0, T*B) are h_{-1} .. h_{T-2}
            for ki, (w, bias, dp, inv_s, u_, v_, g_, hsrc) in enumerate(((wr, br, dpr, isr, ur, vr, gr, hprev_all),
                                                                         (wu, bu, dpu, isu, uu, vu, gu, hprev_all),
                                                                         (wc, bc, dpc, isc, uc, vc, gc, rh))):
                m = tb * hh * ww
                want_bias = bias is not None and bias.requires_grad
                if not w.requires_grad:
                    if want_bias:
                        tmpd = colsum_tmp(m, ch, dev)
                        call("dgmr_colsum", _p(dp), _p(grad_buffer(bias)), _p(tmpd), m, ch, 1, st)
                    continue
                g = torch.empty(ch * taps * ct, device=dev, dtype=torch.float32)
                dot = dot_buffer(g_, dev)
                # x half: T*B maps, or (shared x) the T per-step sums against T copies of the one map; h half: always T*B maps
                x_half = (x_rep, dsum[ki], TD, cx, 0) if x_shared else (x_all, dp, tb, cx, 0)
                for src, dy, nimg, cin, coff in (x_half, (hsrc, dp, tb, ch, cx)):
                    k = taps * cin
                    wa = WgradArgs()
                    wa.x, wa.dy = _p(src), _p(dy)
                    wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout = nimg, 1, hh, ww, cin, ch
                    wa.KD, wa.KH, wa.KW = 1, kh, kw
                    wa.upsample, wa.pre_relu, wa.pre_group, wa.groups = 0, 0, 1, g_
                    wa.bias_grad = _p(grad_buffer(bias)) if (want_bias and coff == 0) else None  # bias gradient once, with the x half
                    call("dgmr_conv_wgrad_plan", ctypes.byref(wa))
                    ns = wa.nsplit
                    partial = torch.empty(ns * ch * k, device=dev, dtype=torch.float32)
                    wa.partial = _p(partial)
                    rows_ws = bias_rows(wa, dev)
                    call("dgmr_conv_wgrad", ctypes.byref(wa), st)
                    del rows_ws
                    call("dgmr_wgrad_reduce_slice", _p(partial), ns, g_, ch, taps, cin, ct, coff, _p(w), _p(inv_s), _p(g), _p(dot), st)
                call("dgmr_sn_wgrad_finalize", _p(g), _p(grad_buffer(w)), _p(dot), _p(inv_s), _p(u_), _p(v_), ch, ct, taps, g_, 1, st)

        if _WGRAD_STREAM:
            keep = [buf, rh, dpr, dpu, dpc, x_all, isr, ur, vr, isu, uu, vu, isc, uc, vc]
            if x_shared:
                keep += [x_rep] + dsum
            _on_side_stream(dev, weight_grads, keep, cost=float(tb * hh * ww) * kh * kw * (cx + ch) * 3 * ch)
        else:
            weight_grads()
        # ---- x-part data gradient, batched over the T steps (each step with its own 1/sigma) ----
        dx_all = None
        if ctx.needs_input_grad[0] and x_shared:
            # dx[d] = sum_k convT_k( sum_t dsum_k[t][d] / sigma_k[t][d] ): three convs of ONE map per draw
            dx_all = empty_cl((draws, cx, hh, ww), dout_all)
            tmp = empty_cl((draws, cx, hh, ww), dout_all)
            chain = ((dsum[0], wr, isr, gr, None, tmp), (dsum[1], wu, isu, gu, tmp, dx_all), (dsum[2], wc, isc, gc, dx_all, tmp))
            for ds, w, inv_s, g_, res, dst in chain:
                wsum = empty_cl((draws, ch, hh, ww), dout_all)
                # train: one 1/sigma per (step, draw) = weight [row t][column block d]; eval: a single 1/sigma
                call("dgmr_group_rowsum", _p(ds), _p(inv_s), _p(wsum), 1, T, draws * n_img, 1 if g_ > 1 else T,
                     draws if g_ > 1 else 1, st)
                _launch_conv(wsum, _p(_flipped_weight(w, 0, cx)), None, None, dst, draws, 1, hh, ww, ch, cx, 1, kh, kw, residual=res,
                             w_split=_split_planes(w, True, 0, cx))
            dx_all = tmp
        elif ctx.needs_input_grad[0]:
            dx_all = empty_cl((tb, cx, hh, ww), dout_all)
            tmp = empty_cl((tb, cx, hh, ww), dout_all)
            chain = ((dpr, wr, isr, gr, None, tmp), (dpu, wu, isu, gu, tmp, dx_all), (dpc, wc, isc, gc, dx_all, tmp))
            for dp, w, inv_s, g_, res, dst in chain:
                _launch_conv(dp, _p(_flipped_weight(w, 0, cx)), None, inv_s, dst, tb, 1, hh, ww, ch, cx, 1, kh, kw, residual=res,
                             scale_group=tb // g_, w_split=_split_planes(w, True, 0, cx))
            dx_all = tmp
        return dx_all, dh0, None, None, None, None, None, None


def conv_gru(x_all, h0, params, seqs, steps: int, x_shared: bool = False, draws: int = 1):
    return ConvGRUFn.apply(x_all, h0, params, seqs, steps, x_shared, draws, params[0])
