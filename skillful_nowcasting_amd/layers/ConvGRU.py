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

    def forward_batched(self, x_all: torch.Tensor, hidden_state: torch.Tensor, steps: int, x_shared: bool = False, draws: int = 1,
                        layout=None) -> torch.Tensor:
        """All `steps` inputs as one time-major batch [steps*B, C, h, w] -> all outputs [steps*B, C_out, h, w].

        x_shared: x_all is a single map [1, C, h, w] fed to every sample at every step (the sampler's `[latent] * T`).
        draws > 1: B = draws * B' samples per step, draw-major; `layout` = CallLayout(draws, steps, time_major=True, ...) tells
        which of the steps * draws calls of the reference each (step, draw) group is; a shared x is one map per draw."""
        cell = self.cell
        convs = (cell.read_gate_conv, cell.update_gate_conv, cell.output_conv)
        # the spectral-norm iterations of the three convs do not depend on the data: all calls are drawn up front
        seqs = tuple(c._sigma(steps * draws, layout) for c in convs)
        params = tuple(p for c in convs for p in (c.weight_orig, c.bias))
        return ops.conv_gru(x_all, hidden_state, params, seqs, steps, x_shared, draws)

    def forward_list(self, x, hidden_state=None) -> List[torch.Tensor]:
        steps = len(x)
        return ops.unstack_batch(self.forward_batched(ops.stack_batch(list(x)), hidden_state, steps), steps)

    def forward(self, x, hidden_state=None) -> torch.Tensor:
        return torch.stack(self.forward_list(x, hidden_state), dim=0)
