"""ConvGRU and cell (mirror of dgmr/layers/ConvGRU.py) on the HIP operators."""
from typing import List

import torch

from .. import ops
from ..nn import SNConv


class ConvGRUCell(torch.nn.Module):
    """dgmr/layers/ConvGRU.py:8-85: r, u = sigmoid(SN-conv3x3([x, h])); c = relu(SN-conv3x3([x, r*h])); h' = u*h + (1-u)*c.

    The three convs write pre-activations; sigmoid/relu and the products run in two fused gating kernels.
    """

    def __init__(self, input_channels: int, output_channels: int, kernel_size: int = 3, sn_eps: float = 0.0001):
        super().__init__()
        self._kernel_size = kernel_size
        self._sn_eps = sn_eps
        self.read_gate_conv = SNConv(input_channels, output_channels, kernel_size, eps=sn_eps)
        self.update_gate_conv = SNConv(input_channels, output_channels, kernel_size, eps=sn_eps)
        self.output_conv = SNConv(input_channels, output_channels, kernel_size, eps=sn_eps)

    def forward(self, x, prev_state, sn=None):
        """`sn`: optional (read, update, output) spectral-norm records drawn up front for this step (ConvGRU.forward_list)."""
        sr, su, so = sn if sn is not None else (None, None, None)
        xh = ops.cat_channels([x, prev_state])
        pre_read = self.read_gate_conv(xh, sn=sr)
        pre_update = self.update_gate_conv(xh, sn=su)
        gated_input = ops.cat_channels([x, ops.gru_gate(pre_read, prev_state)])
        pre_c = self.output_conv(gated_input, sn=so)
        out = ops.gru_blend(pre_update, prev_state, pre_c)
        return out, out


class ConvGRU(torch.nn.Module):
    """dgmr/layers/ConvGRU.py:88-111."""

    def __init__(self, input_channels: int, output_channels: int, kernel_size: int = 3, sn_eps=0.0001):
        super().__init__()
        self.cell = ConvGRUCell(input_channels, output_channels, kernel_size, sn_eps)

    def forward_list(self, x, hidden_state=None) -> List[torch.Tensor]:
        outputs = []
        steps = len(x)
        # the spectral-norm iterations of the cell's three convs do not depend on the data: all `steps` of them are drawn up front
        # in one pass per conv (dgmr_spectral_sigma_seq), then step t uses record t
        cell = self.cell
        seqs = [c._sigma(steps) for c in (cell.read_gate_conv, cell.update_gate_conv, cell.output_conv)]
        for step in range(steps):
            # groups == 1: a single step, or eval mode (no iteration: every step sees the same sigma)
            sn = tuple(q.at(step) if q.groups > 1 else q for q in seqs)
            output, hidden_state = cell(x[step], hidden_state, sn)
            outputs.append(output)
        return outputs

    def forward(self, x, hidden_state=None) -> torch.Tensor:
        return torch.stack(self.forward_list(x, hidden_state), dim=0)
