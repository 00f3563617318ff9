"""Latent self-attention (mirror of dgmr/layers/Attention.py) on the HIP operators."""
import torch
import torch.nn as nn

from .. import ops
from ..nn import Conv


class AttentionLayer(torch.nn.Module):
    """dgmr/layers/Attention.py:23-85.  q/k/v/last are bias-free 1x1 convs; gamma is the learnable gain.

    The attention product reproduces the reference's einsum over the NCHW view (see dgmr_attention_fwd in
    include/dgmr_hip.h); ``gamma * last_conv(out) + x`` is one conv launch (epilogue scale + residual).
    """

    def __init__(self, input_channels: int, output_channels: int, ratio_kq: int = 8, ratio_v: int = 8):
        super().__init__()
        self.ratio_kq = ratio_kq
        self.ratio_v = ratio_v
        self.output_channels = output_channels
        self.input_channels = input_channels
        self.query = Conv(input_channels, output_channels // ratio_kq, 1, bias=False)
        self.key = Conv(input_channels, output_channels // ratio_kq, 1, bias=False)
        self.value = Conv(input_channels, output_channels // ratio_v, 1, bias=False)
        self.last_conv = Conv(output_channels // 8, output_channels, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        query = self.query(x)
        key = self.key(x)
        value = self.value(x)
        out = ops.attention(query, key, value)
        return self.last_conv(out, scale=self.gamma, gamma_scale=True, residual=x)
