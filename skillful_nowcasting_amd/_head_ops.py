"""Small operators around the convolutions: latent attention, the discriminator heads (relu-sum, BatchNorm1d, spectral-norm
linear), the losses, device-side scalar bookkeeping and the Adam update - autograd Functions over the C ABI."""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from ._lib import ConvArgs, WgradArgs, call
from ._core import (BNState, SNCall, _copy, _dims, _p, _stream, bn_prepare, dot_buffer, empty_cl, grad_buffer, require_hip, sums_buffer, colsum_tmp,
                    to_cl)


# ---------------------------------------------------------------------------------------------------
# latent attention (dgmr/layers/Attention.py:9-20,78-82)
# ---------------------------------------------------------------------------------------------------
class AttentionFn(Function):
    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = to_cl(q), to_cl(k), to_cl(v)
        require_hip(q)
        b, cq, h, w = q.shape
        if v.shape[1] != cq:
            raise RuntimeError("attention: ratio_kq must equal ratio_v (the reference's einsum requires it)")
        L = cq * h
        out = torch.empty_like(v)
        beta = torch.empty(b, L, L, device=q.device, dtype=torch.float32)
        n = cq * h * w
        for i in range(b):
            o = 4 * n * i
            call("dgmr_attention_fwd", q.data_ptr() + o, k.data_ptr() + o, v.data_ptr() + o, beta.data_ptr() + 4 * L * L * i,
                 out.data_ptr() + o, cq, h, w, _stream())
        ctx.save_for_backward(q, k, v, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, beta = ctx.saved_tensors
        dout = to_cl(dout)
        b, cq, h, w = q.shape
        L = cq * h
        n = cq * h * w
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        tmp = torch.empty(L * L, device=q.device, dtype=torch.float32)
        for i in range(b):
            o = 4 * n * i
            call("dgmr_attention_bwd", dout.data_ptr() + o, q.data_ptr() + o, k.data_ptr() + o, v.data_ptr() + o,
                 beta.data_ptr() + 4 * L * L * i, dq.data_ptr() + o, dk.data_ptr() + o, dv.data_ptr() + o, _p(tmp), cq, h, w,
                 _stream())
        return dq, dk, dv


attention = AttentionFn.apply


# ---------------------------------------------------------------------------------------------------
# discriminator heads (discriminators.py:127-131,217-219)
# ---------------------------------------------------------------------------------------------------
class ReluSumHWFn(Function):
    @staticmethod
    def forward(ctx, x):
        require_hip(x)
        x = to_cl(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c, device=x.device, dtype=torch.float32)
        call("dgmr_relu_sum_hw_fwd", _p(x), _p(y), n, h * w, c, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        call("dgmr_relu_sum_hw_bwd", _p(dy.contiguous()), _p(x), _p(dx), n, h * w, c, _stream())
        return dx


relu_sum_hw = ReluSumHWFn.apply


class BatchNorm1dFn(Function):
    """torch.nn.BatchNorm1d on [N, C] (discriminators.py:102,129,194,218), batch statistics in train mode."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, eps, momentum, train, groups=1, layout=None):
        require_hip(x)
        x = x.contiguous()
        n, c = x.shape
        st = bn_prepare(x.view(n, c, 1, 1), gamma, beta, running_mean, running_var, nbt, eps, momentum, train, groups, layout)
        y = torch.empty_like(x)
        if st.groups > 1:
            call("dgmr_affine", _p(x), _p(st.a), _p(st.b), _p(y), st.groups, n // st.groups, c, 0, _stream())
        else:
            call("dgmr_affine", _p(x), _p(st.a), _p(st.b), _p(y), 1, n, c, 0, _stream())
        ctx.st = st
        ctx.save_for_backward(x, st.mean, st.rstd)  # see ConvFn.forward: nothing tensor-valued may be read from ctx.st
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        st: BNState = ctx.st
        n, c = x.shape
        dy = dy.contiguous()
        gq = st.groups
        sums = sums_buffer(gq, n // gq, c, x.device)
        call("dgmr_bn_bwd_reduce", _p(dy), _p(x), _p(mean), _p(rstd), _p(sums), gq, n // gq, c, _stream())
        dx = torch.empty_like(x)
        dgam = grad_buffer(st.gamma) if st.gamma.requires_grad else None
        dbet = grad_buffer(st.beta) if st.beta.requires_grad else None
        call("dgmr_bn_bwd_apply", _p(dy), _p(x), _p(mean), _p(rstd), _p(st.gamma), _p(sums), None, _p(dx), _p(dgam), _p(dbet),
             gq, n // gq, c, int(st.train), _stream())
        return dx, None, None, None, None, None, None, None, None, None, None


class SNLinear1Fn(Function):
    """spectral_norm(Linear(C, 1)) (discriminators.py:100,192); `sn.groups` calls (frames) per launch."""

    @staticmethod
    def forward(ctx, x, w, bias, sn: SNCall):
        require_hip(x)
        x = x.contiguous()
        n, c = x.shape
        if n % sn.groups:
            raise RuntimeError(f"linear: {n} rows are not divisible into {sn.groups} spectral-norm call groups")
        y = torch.empty(n, 1, device=x.device, dtype=torch.float32)
        call("dgmr_linear1_fwd", _p(x), _p(w), _p(bias), _p(sn.inv_sigma), _p(y), n, c, n // sn.groups, _stream())
        ctx.groups = sn.groups
        ctx.params = (w, bias)
        ctx.save_for_backward(x, sn.inv_sigma, sn.u, sn.v)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, inv_sigma, sn_u, sn_v = ctx.saved_tensors
        w, bias = ctx.params
        n, c = x.shape
        gq = ctx.groups
        dy = dy.contiguous()
        dev = x.device
        dx = torch.empty_like(x)
        g = torch.empty(gq * c, device=dev, dtype=torch.float32)
        gb = torch.empty(1, device=dev, dtype=torch.float32)
        st = _stream()
        call("dgmr_linear1_bwd", _p(dy), _p(x), _p(w), _p(inv_sigma), _p(dx), _p(g), _p(gb), n, c, n // gq, st)
        if bias is not None and bias.requires_grad:
            b = grad_buffer(bias)
            call("dgmr_axpby", _p(b), _p(gb), _p(b), 1.0, 1.0, 1, st)
        if w.requires_grad:
            dot = dot_buffer(gq, dev)
            g2 = torch.empty(c, device=dev, dtype=torch.float32)
            call("dgmr_wgrad_reduce", _p(g), gq, gq, c, _p(w), _p(inv_sigma), _p(g2), _p(dot), st)
            call("dgmr_sn_wgrad_finalize", _p(g2), _p(grad_buffer(w)), _p(dot), _p(inv_sigma), _p(sn_u), _p(sn_v), 1, c, 1, gq, 1, st)
        return dx, None, None, None


# ---------------------------------------------------------------------------------------------------
# losses (dgmr/losses.py:172-192,307-319 ; dgmr/dgmr.py:20-33)
# ---------------------------------------------------------------------------------------------------
class HingeDiscFn(Function):
    @staticmethod
    def forward(ctx, score_generated, score_real):
        require_hip(score_real)
        sg, sr = score_generated.contiguous(), score_real.contiguous()
        loss = torch.empty((), device=sr.device, dtype=torch.float32)
        dg, dr = torch.empty_like(sg), torch.empty_like(sr)
        call("dgmr_hinge_disc", _p(sr), _p(sg), _p(loss), _p(dr), _p(dg), sr.numel(), sg.numel(), _stream())
        ctx.save_for_backward(dg, dr)
        return loss

    @staticmethod
    def backward(ctx, gl):
        dg, dr = ctx.saved_tensors
        gl = gl.contiguous()
        og, orr = torch.empty_like(dg), torch.empty_like(dr)
        call("dgmr_scale_by_dev", _p(dg), _p(gl), 1.0, _p(og), dg.numel(), _stream())
        call("dgmr_scale_by_dev", _p(dr), _p(gl), 1.0, _p(orr), dr.numel(), _stream())
        return og, orr


class MeanFn(Function):
    """sign * mean(x) (loss_hinge_gen = -mean)."""

    @staticmethod
    def forward(ctx, x, sign: float):
        require_hip(x)
        x = x.contiguous()
        n = x.numel()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        tmp = colsum_tmp(n, 1, x.device)
        call("dgmr_colsum", _p(x), _p(out), _p(tmp), n, 1, 0, _stream())
        call("dgmr_axpby", _p(out), None, _p(out), sign / n, 0.0, 1, _stream())
        ctx.n, ctx.sign, ctx.shape = n, sign, x.shape
        return out

    @staticmethod
    def backward(ctx, gl):
        g = torch.empty(ctx.shape, device=gl.device, dtype=torch.float32)
        ones = torch.ones(ctx.shape, device=gl.device, dtype=torch.float32)
        call("dgmr_scale_by_dev", _p(ones), _p(gl.contiguous()), ctx.sign / ctx.n, _p(g), ctx.n, _stream())
        return g, None


class GridCellFn(Function):
    """GridCellLoss on the mean of K stacked predictions: || (mean_k g_k - y) * max(y+1, cap) ||_1 / T * H * W."""

    @staticmethod
    def forward(ctx, preds, targets, cap: float, weights=None):
        """`weights`: explicit per-element weights (a caller-supplied weight_fn evaluated on the targets); None = the reference's
        default max(y + 1, cap), evaluated inside the kernel."""
        require_hip(preds)
        preds, targets = preds.contiguous(), targets.contiguous()
        if weights is not None:
            weights = weights.expand_as(targets).contiguous().float()
        k = preds.shape[0]
        n = targets.numel()
        mult = float(targets.size(3) * targets.size(4)) / float(targets.size(1))
        loss = torch.empty((), device=preds.device, dtype=torch.float32)
        from ._lib import load

        acc = torch.zeros(int(load().dgmr_grid_cell_acc_doubles(n)), device=preds.device, dtype=torch.float64)
        dweight = torch.empty_like(targets)
        call("dgmr_grid_cell_loss", _p(preds), k, n, _p(targets), _p(weights), float(cap), _p(acc), _p(loss), mult, _p(dweight), n,
             _stream())
        ctx.save_for_backward(dweight)
        ctx.k, ctx.mult = k, mult
        return loss

    @staticmethod
    def backward(ctx, gl):
        (dweight,) = ctx.saved_tensors
        n = dweight.numel()
        g1 = torch.empty_like(dweight)
        call("dgmr_scale_by_dev", _p(dweight), _p(gl.contiguous()), ctx.mult, _p(g1), n, _stream())
        return g1.unsqueeze(0).expand(ctx.k, *dweight.shape), None, None, None


class AxpbyFn(Function):
    """alpha*a + beta*b on device (loss bookkeeping without torch arithmetic kernels)."""

    @staticmethod
    def forward(ctx, a, b, alpha: float, beta: float):
        require_hip(a)
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        call("dgmr_axpby", _p(a), _p(b), _p(out), alpha, beta, a.numel(), _stream())
        ctx.ab = (alpha, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        alpha, beta = ctx.ab
        g = g.contiguous()
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        call("dgmr_axpby", _p(g), None, _p(ga), alpha, 0.0, g.numel(), _stream())
        call("dgmr_axpby", _p(g), None, _p(gb), beta, 0.0, g.numel(), _stream())
        return ga, gb, None, None


def axpby(a, b, alpha=1.0, beta=1.0):
    return AxpbyFn.apply(a, b, float(alpha), float(beta))


# ---------------------------------------------------------------------------------------------------
# Adam (dgmr/dgmr.py:292-300)
# ---------------------------------------------------------------------------------------------------
def adam_update(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    call("dgmr_adam", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step), _stream())

