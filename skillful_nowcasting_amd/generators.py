"""Sampler and Generator (mirror of dgmr/generators.py) on the HIP operators."""
from typing import List

import torch
from huggingface_hub import PyTorchModelHubMixin

from . import ops
from .common import GBlock, UpsampleGBlock
from .layers import ConvGRU
from .nn import BatchNorm, SNConv, SNScope


class Sampler(torch.nn.Module, PyTorchModelHubMixin):
    """dgmr/generators.py:20-182: 4 x (ConvGRU -> SN-1x1 -> GBlock -> UpsampleGBlock), BN+ReLU, SN-1x1, depth-to-space."""

    def __init__(self, forecast_steps: int = 18, latent_channels: int = 768, context_channels: int = 384,
                 output_channels: int = 1):
        super().__init__()
        self.forecast_steps = forecast_steps
        lc, cc = latent_channels, context_channels
        self.convGRU1 = ConvGRU(lc + cc, cc, 3)
        self.gru_conv_1x1 = SNConv(cc, lc, 1)
        self.g1 = GBlock(lc, lc)
        self.up_g1 = UpsampleGBlock(lc, lc // 2)
        self.convGRU2 = ConvGRU(lc // 2 + cc // 2, cc // 2, 3)
        self.gru_conv_1x1_2 = SNConv(cc // 2, lc // 2, 1)
        self.g2 = GBlock(lc // 2, lc // 2)
        self.up_g2 = UpsampleGBlock(lc // 2, lc // 4)
        self.convGRU3 = ConvGRU(lc // 4 + cc // 4, cc // 4, 3)
        self.gru_conv_1x1_3 = SNConv(cc // 4, lc // 4, 1)
        self.g3 = GBlock(lc // 4, lc // 4)
        self.up_g3 = UpsampleGBlock(lc // 4, lc // 8)
        self.convGRU4 = ConvGRU(lc // 8 + cc // 8, cc // 8, 3)
        self.gru_conv_1x1_4 = SNConv(cc // 8, lc // 8, 1)
        self.g4 = GBlock(lc // 8, lc // 8)
        self.up_g4 = UpsampleGBlock(lc // 8, lc // 16)
        self.bn = BatchNorm(lc // 16)
        self.relu = torch.nn.ReLU()
        self.conv_1x1 = SNConv(lc // 16, 4 * output_channels, 1)
        self.depth2space = torch.nn.PixelShuffle(upscale_factor=2)

    def forward(self, conditioning_states: List[torch.Tensor], latent_dim: torch.Tensor, draws: int = 1,
                reverse: bool = False) -> torch.Tensor:
        """`draws` > 1: `draws` generator forwards of the reference in one go - conditioning states [draws * B, ...] (draw-major),
        one latent map per draw [draws, C, h, w]; returns [draws * B, T, C, H, W].  Every (draw, step) keeps its own spectral-norm
        sigma and BatchNorm statistics, advanced in the reference's order (draw-major; `reverse`: last draw first)."""
        init_states = conditioning_states
        T = self.forecast_steps
        lay = ops.CallLayout(draws, T, time_major=True, reverse=reverse) if (draws > 1 or reverse) else None
        calls = T * draws
        # `[repeat(latent, B)] * T` (generators.py:146-149): the first ConvGRU sees the same map for every sample at every step,
        # so it is handed over as the single map it is (ConvGRUFn x_shared) instead of T*B copies
        h = latent_dim
        levels = ((self.convGRU1, self.gru_conv_1x1, self.g1, self.up_g1),
                  (self.convGRU2, self.gru_conv_1x1_2, self.g2, self.up_g2),
                  (self.convGRU3, self.gru_conv_1x1_3, self.g3, self.up_g3),
                  (self.convGRU4, self.gru_conv_1x1_4, self.g4, self.up_g4))
        for lvl, (gru, c11, g, upg) in enumerate(levels):
            # only the ConvGRU is a true recurrence; its T outputs then travel as ONE time-major batch [T*B, C, h, w] through
            # the 1x1 conv, the G-block and the upsampling G-block (one launch per conv instead of T), every forecast step
            # keeping its own spectral-norm sigma and BatchNorm batch statistics exactly as the reference's T calls do
            h = gru.forward_batched(h, init_states[3 - lvl], T, x_shared=(lvl == 0), draws=draws, layout=lay)
            h = c11(h, calls=calls, layout=lay)
            # BatchNorm batch statistics travel with the tensors: the conv that writes a BatchNorm's input sums it in its epilogue
            h, st = g(h, calls=calls, layout=lay, out_stats=True)
            if lvl == 3:
                h, st = upg(h, calls=calls, layout=lay, in_stats=st, out_stats=True)
            else:
                h, st = upg(h, calls=calls, layout=lay, in_stats=st), None
        # relu(bn(h)) folded into the 1x1 conv's operand load; PixelShuffle + stack in one layout pass
        h = self.conv_1x1(h, bn=self.bn.prepare(h, calls, lay, st), calls=calls, layout=lay)
        return ops.d2s_frames(h, T)


class Generator(torch.nn.Module, PyTorchModelHubMixin):
    """dgmr/generators.py:185-212."""

    def __init__(self, conditioning_stack: torch.nn.Module, latent_stack: torch.nn.Module, sampler: torch.nn.Module):
        super().__init__()
        self.conditioning_stack = conditioning_stack
        self.latent_stack = latent_stack
        self.sampler = sampler

    def forward(self, x: torch.Tensor):
        return self.forward_draws(x, 1)

    def forward_draws(self, x: torch.Tensor, draws: int, reverse: bool = False, zs=None) -> torch.Tensor:
        """`draws` forward calls of the reference on the same frames x as ONE batch -> [draws * B, T, C, H, W], draw-major.

        Train mode: identical to calling `forward(x)` `draws` times in a row (each call draws its own latent z from the CPU
        generator in that order, and advances every spectral-norm u / v and BatchNorm running statistic once per call of the
        module) - the per-step kernels just see `draws` times more rows.  `reverse=True` assigns the state sequence as if the
        calls had been made last-draw-first, which is what activation checkpointing's recompute does (dgmr/dgmr.py:176).
        Eval mode: ensemble sampling - the context stack runs once (its result does not depend on the draw), the sampler on all draws.
        `zs`: latent draws to use ([draws, 8, h/32, w/32]) instead of drawing them here.
        """
        if zs is None:
            zs = torch.cat([self.latent_stack.draw(x) for _ in range(draws)], dim=0)
        elif zs.shape[0] != draws:
            raise RuntimeError(f"forward_draws: {zs.shape[0]} latent draws given for draws={draws}")
        # every spectral-norm power iteration of this forward is data-independent: after the first (traced) call they are all
        # drawn up front in three launches
        with SNScope(self, (tuple(x.shape), getattr(self.sampler, "forecast_steps", 0), draws, reverse)):
            per_call = self.training and (draws > 1 or reverse)
            if per_call:
                conditioning_states = self.conditioning_stack(x, draws=draws, reverse=reverse)
            else:
                conditioning_states = self.conditioning_stack(x)
                if draws > 1:
                    conditioning_states = [ops.repeat_batch(c, draws) for c in conditioning_states]
            latent_dim = self.latent_stack.forward_latent(zs, reverse=reverse and self.training)
            return self.sampler(conditioning_states, latent_dim, draws=draws, reverse=reverse and self.training)
