"""Parameter containers with the reference's ``state_dict`` layout, computing through the HIP operators.

``SNConv`` / ``SNLinear1`` reproduce the key set that ``torch.nn.utils.parametrizations.spectral_norm``
gives a Conv/Linear (``bias``, ``parametrizations.weight.original``, ``parametrizations.weight.0._u``,
``parametrizations.weight.0._v``) and the same construction-time RNG consumption (kaiming-uniform weight,
uniform bias, normal u/v, 15 power iterations — torch/nn/modules/conv.py reset_parameters and
torch/nn/utils/parametrizations.py:432-439), so a seeded construction yields the reference's tensors.
Construction runs on the host with torch's init functions; every forward/backward op is a HIP kernel.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .ops import BNState, ConvSpec


def _mf(ndim: int):
    return torch.channels_last if ndim == 4 else (torch.channels_last_3d if ndim == 5 else torch.contiguous_format)


def _conv_init(out_channels: int, in_channels: int, ks):
    w = torch.empty(out_channels, in_channels, *ks)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    fan_in = in_channels * math.prod(ks)
    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
    b = torch.empty(out_channels)
    nn.init.uniform_(b, -bound, bound)
    return w, b


class _SNVectors(nn.Module):
    def __init__(self, u, v):
        super().__init__()
        self.register_buffer("_u", u)
        self.register_buffer("_v", v)


class _SNWeight(nn.Module):
    def __init__(self, weight, u, v):
        super().__init__()
        self.original = nn.Parameter(weight)
        self.add_module("0", _SNVectors(u, v))


class _Parametrizations(nn.Module):
    def __init__(self, snw):
        super().__init__()
        self.weight = snw


class _SpectralNormBase(nn.Module):
    """Holds W (``parametrizations.weight.original``), bias, u, v; one power iteration per train-mode call."""

    def _init_sn(self, w: torch.Tensor, b: torch.Tensor, eps: float):
        self.eps = eps
        self.bias = nn.Parameter(b)
        wm = w.flatten(1)
        h, wd = wm.shape
        u = F.normalize(wm.new_empty(h).normal_(0, 1), dim=0, eps=eps)
        v = F.normalize(wm.new_empty(wd).normal_(0, 1), dim=0, eps=eps)
        with torch.no_grad():
            # 15 warm-up iterations (parametrizations.py:432-439) + the train-mode forward that
            # register_parametrization runs once as its consistency check = 16 before the first user call
            for _ in range(16):
                u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
                v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
        self.parametrizations = _Parametrizations(_SNWeight(w.contiguous(memory_format=_mf(w.dim())), u, v))
        self.register_buffer("_scratch", torch.zeros(4), persistent=False)

    @property
    def weight_orig(self) -> torch.Tensor:
        return self.parametrizations.weight.original

    def _gram(self) -> torch.Tensor:
        """W W^T, recomputed only when the optimiser (or a load_state_dict) has changed W."""
        w = self.weight_orig
        tag = (w._version, ops.weights_epoch(), w.data_ptr())
        hit = getattr(self, "_gram_cache", None)
        if hit is None or hit[0] != tag:
            hit = (tag, ops.weight_gram(w))
            self._gram_cache = hit
        return hit[1]

    def _sigma(self, calls: int = 1) -> ops.SNCall:
        """Spectral-norm record of `calls` consecutive calls of this module (train: one power iteration per call)."""
        vec = getattr(self.parametrizations.weight, "0")
        if calls > 1 and self.training:
            return ops.spectral_sigma_seq(self.weight_orig, self._gram(), vec._u, vec._v, self._scratch, self.eps, calls)
        # one call, or eval mode (no iteration: every call sees the same sigma)
        return ops.spectral_sigma(self.weight_orig, vec._u, vec._v, self._scratch, self.eps, self.training)


class SNConv(_SpectralNormBase):
    """spectral_norm(Conv2d/Conv3d(k in {1,3}, stride 1, 'same' padding))."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, ndim: int = 2, eps: float = 1e-12):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.ndim = in_channels, out_channels, kernel_size, ndim
        w, b = _conv_init(out_channels, in_channels, (kernel_size,) * ndim)
        self._init_sn(w, b, eps)

    def forward(self, x, *, pre_relu: bool = False, bn: Optional[BNState] = None, upsample: bool = False, residual=None,
                act_relu: bool = False, calls: int = 1, sn: Optional[ops.SNCall] = None):
        """`calls` > 1: x is a batch of `calls` groups (forecast steps / frames), each group being one call of this module in
        the reference (own power iteration, own sigma).  `sn`: a record drawn earlier with `_sigma` (ConvGRU steps)."""
        if sn is None:
            sn = self._sigma(calls)
        spec = ConvSpec(upsample=upsample, pre_relu=pre_relu, bn=bn, sn=sn, act_relu=act_relu)
        return ops.conv(x, self.weight_orig, self.bias, sn.inv_sigma, residual, spec)


class SNLinear1(_SpectralNormBase):
    """spectral_norm(Linear(C, 1)) — the discriminator heads (discriminators.py:100,192)."""

    def __init__(self, in_features: int, eps: float = 1e-12):
        super().__init__()
        self.in_features = in_features
        w = torch.empty(1, in_features)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        b = torch.empty(1)
        nn.init.uniform_(b, -bound, bound)
        self._init_sn(w, b, eps)

    def forward(self, x, calls: int = 1):
        sn = self._sigma(calls)
        return ops.SNLinear1Fn.apply(x, self.weight_orig, self.bias, sn)


class Conv(nn.Module):
    """Plain Conv2d(k in {1,3}, 'same' padding) with keys ``weight`` / ``bias`` (LBlock, Attention)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, bias: bool = True):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        w, b = _conv_init(out_channels, in_channels, (kernel_size, kernel_size)) if bias else (None, None)
        if not bias:
            w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        if bias:
            self.bias = nn.Parameter(b)
        else:
            self.register_parameter("bias", None)

    def forward(self, x, *, pre_relu: bool = False, residual=None, scale=None, gamma_scale: bool = False):
        spec = ConvSpec(pre_relu=pre_relu, gamma_scale=gamma_scale)
        return ops.conv(x, self.weight, self.bias, scale, residual, spec)


class BatchNorm(nn.BatchNorm2d):
    """Parameter/buffer container with BatchNorm2d's keys and init; never run through torch's batch_norm.

    ``prepare(x)`` launches the statistics kernels and returns the per-channel affine that the NEXT conv
    applies (with the ReLU) while loading its operand: the normalised tensor is never written to HBM.
    """

    def forward(self, x):  # pragma: no cover - guard
        raise RuntimeError("BatchNorm is fused into the following conv; call .prepare(x) and pass bn= to the conv")

    def prepare(self, x, groups: int = 1) -> BNState:
        return ops.bn_prepare(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked, self.eps,
                              self.momentum, self.training, groups)


class BatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d over [N, C] through the HIP kernels (discriminator heads)."""

    def forward(self, x, groups: int = 1):
        """`groups` > 1: x is [groups*N, C]; each group is one call of the reference's module (own batch statistics)."""
        return ops.BatchNorm1dFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                                       self.eps, self.momentum, self.training, groups)
