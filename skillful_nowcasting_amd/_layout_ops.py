"""Layout moves of the DGMR step as autograd Functions over the C ABI: pooling, space-to-depth / depth-to-space of frame stacks,
channel concatenation, batch <-> list and time <-> channel moves for the time-batched modules.  HBM-bound copies, no arithmetic
beyond the 2x2 / 2x2x2 averages."""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import ConvArgs, WgradArgs, call
from ._core import _copy, _dims, _p, _stream, colsum_tmp, empty_cl, require_hip, sums_buffer, to_cl


# ---------------------------------------------------------------------------------------------------
# pooling / layout
# ---------------------------------------------------------------------------------------------------
class PoolAddFn(Function):
    """AvgPool2d(2) / AvgPool3d(2) (+ addend): dgmr/common.py:189-191,225,236-237."""

    @staticmethod
    def forward(ctx, x, addend, pd: int):
        require_hip(x)
        x = to_cl(x)
        n, c, d, h, w = _dims(x)
        oshape = (n, c, h // 2, w // 2) if x.dim() == 4 else (n, c, d // pd, h // 2, w // 2)
        y = empty_cl(oshape, x)
        if addend is not None:
            addend = to_cl(addend)
        call("dgmr_pool_fwd", _p(x), _p(addend), _p(y), n, d, h, w, c, pd, 0.0, None, None, None, 1, _stream())
        ctx.geom = (n, c, d, h, w, pd, x.dim())
        ctx.has_addend = addend is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, d, h, w, pd, nd = ctx.geom
        dy = to_cl(dy)
        dx = empty_cl((n, c, h, w) if nd == 4 else (n, c, d, h, w), dy)
        call("dgmr_pool_bwd", _p(dy), _p(dx), n, d, h, w, c, pd, 0.0, _stream())
        return dx, (dy if ctx.has_addend else None), None


def avg_pool_add(x, addend=None, pd: int = 1):
    return PoolAddFn.apply(x, addend, pd)


class FramesS2DFn(Function):
    """[B,T,C,H,W] frames -> (optional AvgPool2d(2)) -> PixelUnshuffle(2) -> channels-last batch of frames.

    discriminators.py:106-108,202-203 ; common.py:393,400.
    """

    @staticmethod
    def forward(ctx, frames, idx, pool: bool, frame_major: bool, as_3d: bool, idx_group: int = 0):
        require_hip(frames)
        frames = frames.contiguous()
        b, t, c, h, w = frames.shape
        if idx is not None and idx.dim() == 2:  # [calls][F]: one row of frame indices per group of idx_group samples
            if idx_group < 1 or b % idx_group or idx.shape[0] != b // idx_group:
                raise RuntimeError(f"frames_s2d: {tuple(idx.shape)} index rows do not fit {b} samples in groups of {idx_group}")
            f = idx.shape[1]
        else:
            idx_group = 0
            f = t if idx is None else idx.numel()
        p = 2 if pool else 1
        ho, wo = h // (2 * p), w // (2 * p)
        if as_3d:
            out = empty_cl((b, 4 * c, f, ho, wo), frames)
        else:
            out = empty_cl((b * f, 4 * c, ho, wo), frames)
        call("dgmr_frames_s2d", _p(frames), _p(idx), _p(out), b, t, c, h, w, f, int(pool), int(frame_major), idx_group, _stream())
        ctx.geom = (b, t, c, h, w, f, int(pool), int(frame_major), idx_group)
        ctx.idx = idx
        return out

    @staticmethod
    def backward(ctx, dout):
        b, t, c, h, w, f, pool, fm, idx_group = ctx.geom
        dout = to_cl(dout)
        dfr = torch.empty(b, t, c, h, w, device=dout.device, dtype=torch.float32)  # (the kernel writes every element)
        call("dgmr_frames_s2d_bwd", _p(dout), _p(ctx.idx), _p(dfr), b, t, c, h, w, f, pool, fm, idx_group, _stream())
        return dfr, None, None, None, None, None


def frames_s2d(frames, idx=None, pool=False, frame_major=False, as_3d=False, idx_group: int = 0):
    return FramesS2DFn.apply(frames, idx, pool, frame_major, as_3d, idx_group)


class D2SFramesFn(Function):
    """T channels-last maps [B,4C,h,w] -> PixelShuffle(2) -> stacked frames [B,T,C,2h,2w] (generators.py:178-181)."""

    @staticmethod
    def forward(ctx, x, t: int):
        x = to_cl(x)
        require_hip(x)
        tb, c4, h, w = x.shape
        c, b = c4 // 4, tb // t
        frames = torch.empty(b, t, c, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
        n = b * c4 * h * w
        for i in range(t):
            call("dgmr_d2s_frames", x.data_ptr() + 4 * n * i, _p(frames), b, t, i, c, h, w, _stream())
        ctx.geom = (b, t, c, h, w)
        return frames

    @staticmethod
    def backward(ctx, dfr):
        b, t, c, h, w = ctx.geom
        dfr = dfr.contiguous()
        dx = empty_cl((t * b, 4 * c, h, w), dfr)
        n = b * 4 * c * h * w
        for i in range(t):
            call("dgmr_d2s_frames_bwd", _p(dfr), dx.data_ptr() + 4 * n * i, b, t, i, c, h, w, _stream())
        return dx, None


def d2s_frames(x: torch.Tensor, t: int):
    """Time-major batch [T*B, 4C, h, w] -> PixelShuffle(2) -> frames [B, T, C, 2h, 2w]."""
    return D2SFramesFn.apply(x, t)


class CatChannelsFn(Function):
    """torch.cat(dim=1) on channels-last tensors; `interleave`: 'b t c h w -> b (c t) h w' (common.py:423)."""

    @staticmethod
    def forward(ctx, interleave: bool, *xs):
        xs = [to_cl(x) for x in xs]
        require_hip(xs[0])
        cs = [x.shape[1] for x in xs]
        ctot = sum(cs)
        shape = list(xs[0].shape)
        shape[1] = ctot
        out = empty_cl(shape, xs[0])
        r = xs[0].numel() // cs[0]
        off = 0
        for i, x in enumerate(xs):
            if interleave:
                call("dgmr_copy_channels", _p(x), _p(out), r, cs[i], cs[i], 0, 1, ctot, i, len(xs), 0, _stream())
            else:
                call("dgmr_copy_channels", _p(x), _p(out), r, cs[i], cs[i], 0, 1, ctot, off, 1, 0, _stream())
            off += cs[i]
        ctx.cs, ctx.interleave, ctx.r = cs, interleave, r
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = to_cl(dout)
        cs, ctot = ctx.cs, sum(ctx.cs)
        outs = []
        off = 0
        for i, c in enumerate(cs):
            shape = list(dout.shape)
            shape[1] = c
            dx = empty_cl(shape, dout)
            if ctx.interleave:
                call("dgmr_copy_channels", _p(dout), _p(dx), ctx.r, c, ctot, i, len(cs), c, 0, 1, 0, _stream())
            else:
                call("dgmr_copy_channels", _p(dout), _p(dx), ctx.r, c, ctot, off, 1, c, 0, 1, 0, _stream())
            outs.append(dx)
            off += c
        return (None, *outs)


def cat_channels(xs: Sequence[torch.Tensor], interleave: bool = False):
    return CatChannelsFn.apply(interleave, *xs)


class RepeatBatchFn(Function):
    """einops 'b c h w -> (repeat b) c h w' (generators.py:146-148 at b == 1): the whole batch tiled `repeat` times."""

    @staticmethod
    def forward(ctx, x, repeat: int):
        require_hip(x)
        x = to_cl(x)
        out = empty_cl((repeat * x.shape[0],) + tuple(x.shape[1:]), x)
        call("dgmr_repeat_rows", _p(x), _p(out), x.numel(), repeat, _stream())
        ctx.repeat = repeat
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = to_cl(dout)
        n = dout.numel() // ctx.repeat
        dx = empty_cl((dout.shape[0] // ctx.repeat,) + tuple(dout.shape[1:]), dout)
        tmp = colsum_tmp(ctx.repeat, n, dout.device)
        call("dgmr_colsum", _p(dout), _p(dx), _p(tmp), ctx.repeat, n, 0, _stream())
        return dx, None


def repeat_batch(x, repeat: int):
    return RepeatBatchFn.apply(x, repeat)


# ---------------------------------------------------------------------------------------------------
# batch <-> list / time <-> channel layout moves for the T-batched modules
# ---------------------------------------------------------------------------------------------------
class StackBatchFn(Function):
    """T tensors [B, ...] -> one [T*B, ...] (time-major): the per-step outputs of a ConvGRU become ONE batch, so that the
    1x1 / G-block / upsample-G-block convs of all forecast steps run as one launch (generators.py:153-171)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [to_cl(x) for x in xs]
        require_hip(xs[0])
        b = xs[0].shape[0]
        out = empty_cl((b * len(xs),) + tuple(xs[0].shape[1:]), xs[0])
        n = xs[0].numel()
        for i, x in enumerate(xs):
            _copy(_p(x), out.data_ptr() + 4 * n * i, n)
        ctx.t, ctx.b = len(xs), b
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = to_cl(dout)
        n = dout.numel() // ctx.t
        outs = []
        for i in range(ctx.t):
            g = empty_cl((ctx.b,) + tuple(dout.shape[1:]), dout)
            _copy(dout.data_ptr() + 4 * n * i, _p(g), n)
            outs.append(g)
        return tuple(outs)


def stack_batch(xs):
    return StackBatchFn.apply(*xs)


class UnstackBatchFn(Function):
    """[T*B, ...] -> T tensors [B, ...]; the backward writes each gradient into its slot of one buffer."""

    @staticmethod
    def forward(ctx, x, t: int):
        require_hip(x)
        x = to_cl(x)
        b = x.shape[0] // t
        n = x.numel() // t
        outs = []
        for i in range(t):
            o = empty_cl((b,) + tuple(x.shape[1:]), x)
            _copy(x.data_ptr() + 4 * n * i, _p(o), n)
            outs.append(o)
        ctx.t, ctx.shape = t, tuple(x.shape)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        ref = next(d for d in douts if d is not None)
        dx = empty_cl(ctx.shape, ref)
        n = dx.numel() // ctx.t
        for i, d in enumerate(douts):
            if d is None:
                call("dgmr_fill", dx.data_ptr() + 4 * n * i, 0.0, n, _stream())
            else:
                _copy(_p(to_cl(d)), dx.data_ptr() + 4 * n * i, n)
        return dx, None


def unstack_batch(x, t: int):
    return list(UnstackBatchFn.apply(x, t))


class TimeToChannelsFn(Function):
    """[T*B, C, h, w] (time-major batch) -> [B, C*T, h, w] with channel index c*T + t: einops 'b t c h w -> b (c t) h w'
    (common.py:423) read straight from the batched D-block output."""

    @staticmethod
    def forward(ctx, x, t: int, outer: int = 1):
        """outer > 1: x is [outer][T][B] (several generator draws, draw-major) -> [outer * B, C*T, h, w]."""
        require_hip(x)
        x = to_cl(x)
        otb, c, h, w = x.shape
        b = otb // (t * outer)
        out = empty_cl((outer * b, c * t, h, w), x)
        r = b * h * w
        for o in range(outer):
            for i in range(t):
                call("dgmr_copy_channels", x.data_ptr() + 4 * r * c * (o * t + i), out.data_ptr() + 4 * r * c * t * o, r, c, c, 0, 1,
                     c * t, i, t, 0, _stream())
        ctx.geom = (t, b, c, h, w, outer)
        return out

    @staticmethod
    def backward(ctx, dout):
        t, b, c, h, w, outer = ctx.geom
        dout = to_cl(dout)
        dx = empty_cl((outer * t * b, c, h, w), dout)
        r = b * h * w
        for o in range(outer):
            for i in range(t):
                call("dgmr_copy_channels", dout.data_ptr() + 4 * r * c * t * o, dx.data_ptr() + 4 * r * c * (o * t + i), r, c, c * t, i,
                     t, c, 0, 1, 0, _stream())
        return dx, None, None


def time_to_channels(x, t: int, outer: int = 1):
    return TimeToChannelsFn.apply(x, t, outer)


class FramesToBatchFn(Function):
    """[N, C, T, h, w] (channels_last_3d, i.e. N T h w C) -> frame-major batch [T*N, C, h, w]: every `x[:, :, idx]` of the
    temporal discriminator's loop at once (discriminators.py:119-120)."""

    @staticmethod
    def forward(ctx, x):
        require_hip(x)
        x = to_cl(x)
        n, c, t, h, w = x.shape
        out = empty_cl((t * n, c, h, w), x)
        call("dgmr_permute_nt", _p(x), _p(out), n, t, h * w * c, _stream())
        ctx.geom = (n, c, t, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, c, t, h, w = ctx.geom
        dout = to_cl(dout)
        dx = empty_cl((n, c, t, h, w), dout)
        call("dgmr_permute_nt", _p(dout), _p(dx), t, n, h * w * c, _stream())
        return dx


frames_to_batch = FramesToBatchFn.apply


class SumGroupsFn(Function):
    """[G*N, 1] -> [N, 1]: sum over the G frame groups (torch.sum(torch.stack(reps, dim=1), dim=1), discriminators.py:134-137)."""

    @staticmethod
    def forward(ctx, x, groups: int):
        require_hip(x)
        x = x.contiguous()
        n = x.shape[0] // groups
        out = torch.empty(n, 1, device=x.device, dtype=torch.float32)
        tmp = colsum_tmp(groups, n, x.device)
        call("dgmr_colsum", _p(x), _p(out), _p(tmp), groups, n, 0, _stream())
        ctx.groups, ctx.n = groups, n
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dx = torch.empty(ctx.groups * ctx.n, 1, device=dout.device, dtype=torch.float32)
        for i in range(ctx.groups):
            _copy(_p(dout), dx.data_ptr() + 4 * ctx.n * i, ctx.n)
        return dx, None


sum_groups = SumGroupsFn.apply

