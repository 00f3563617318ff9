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

    def forward(self, x, prev_state):
        xh = ops.cat_channels([x, prev_state])
        pre_read = self.read_gate_conv(xh)
        pre_update = self.update_gate_conv(xh)
        gated_input = ops.cat_channels([x, ops.gru_gate(pre_read, prev_state)])
        pre_c = self.output_conv(gated_input)
        out = ops.gru_blend(pre_update, prev_state, pre_c)
        return out, out


class ConvGRU(torch.nn.Module):
    """dgmr/layers/ConvGRU.py:88-111."""

    def __init__(self, input_channels: int, output_channels: int, kernel_size: int = 3, sn_eps=0.0001):
        super().__init__()
        self.cell = ConvGRUCell(input_channels, output_channels, kernel_size, sn_eps)

    def forward_list(self, x, hidden_state=None) -> List[torch.Tensor]:
        outputs = []
        for step in range(len(x)):
            output, hidden_state = self.cell(x[step], hidden_state)
            outputs.append(output)
        return outputs

    def forward(self, x, hidden_state=None) -> torch.Tensor:
        return torch.stack(self.forward_list(x, hidden_state), dim=0)
