"""Generator/discriminator building blocks and conditioning stacks (mirror of dgmr/common.py).

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the reference; the
arithmetic runs in libdgmr_hip.so.  Fusions relative to the reference's op-by-op graph:
  * BatchNorm+ReLU(+nearest-2x upsample) are applied while the following conv loads its operand,
  * every standalone ReLU is a relu-on-load of the consumer conv,
  * spectral-norm's W/sigma is a scalar in the conv epilogue (W/sigma is never materialised),
  * residual adds ride in the conv / pooling epilogues.
"""
from typing import Tuple

import torch
from huggingface_hub import PyTorchModelHubMixin
from torch.distributions import normal

from . import ops
from .layers import AttentionLayer
from .layers.utils import get_conv_layer
from .nn import BatchNorm, Conv, SNConv


class GBlock(torch.nn.Module):
    """Residual generator block without upsampling (dgmr/common.py:17-84)."""

    def __init__(self, input_channels: int = 12, output_channels: int = 12, conv_type: str = "standard",
                 spectral_normalized_eps=0.0001):
        super().__init__()
        self.output_channels = output_channels
        self.bn1 = BatchNorm(input_channels)
        self.bn2 = BatchNorm(input_channels)
        self.relu = torch.nn.ReLU()
        conv2d = get_conv_layer(conv_type)
        self.conv_1x1 = conv2d(input_channels, output_channels, 1, eps=spectral_normalized_eps)
        self.first_conv_3x3 = conv2d(input_channels, input_channels, 3, eps=spectral_normalized_eps)
        self.last_conv_3x3 = conv2d(input_channels, output_channels, 3, eps=spectral_normalized_eps)

    def forward(self, x: torch.Tensor, calls: int = 1, layout=None, in_stats=None, out_stats: bool = False):
        """`calls` > 1: x holds `calls` consecutive calls of this block (forecast steps [x generator draws]) as one batch; every
        call keeps its own BatchNorm batch statistics and spectral-norm sigma; `layout`: ops.CallLayout.
        `in_stats`: partial sums of x taken by the conv that produced it; `out_stats=True`: -> (y, partial sums of y) for the next
        block's first BatchNorm (statistics ride in the producing conv's epilogue instead of a second pass over the tensor)."""
        kw = dict(calls=calls, layout=layout)
        if x.shape[1] != self.output_channels:
            sc = self.conv_1x1(x, **kw)
        else:
            sc = x
        x2, st2 = self.first_conv_3x3(x, bn=self.bn1.prepare(x, calls, layout, in_stats), want_stats=True, **kw)
        return self.last_conv_3x3(x2, bn=self.bn2.prepare(x2, calls, layout, st2), residual=sc, want_stats=out_stats, **kw)


class UpsampleGBlock(torch.nn.Module):
    """Residual generator block with nearest-2x upsampling (dgmr/common.py:87-155)."""

    def __init__(self, input_channels: int = 12, output_channels: int = 12, conv_type: str = "standard",
                 spectral_normalized_eps=0.0001):
        super().__init__()
        self.output_channels = output_channels
        self.bn1 = BatchNorm(input_channels)
        self.bn2 = BatchNorm(input_channels)
        self.relu = torch.nn.ReLU()
        conv2d = get_conv_layer(conv_type)
        self.conv_1x1 = conv2d(input_channels, output_channels, 1, eps=spectral_normalized_eps)
        self.upsample = torch.nn.Upsample(scale_factor=2, mode="nearest")
        self.first_conv_3x3 = conv2d(input_channels, input_channels, 3, eps=spectral_normalized_eps)
        self.last_conv_3x3 = conv2d(input_channels, output_channels, 3, eps=spectral_normalized_eps)

    def forward(self, x: torch.Tensor, calls: int = 1, layout=None, in_stats=None, out_stats: bool = False):
        # shortcut: conv1x1(upsample(x)) == upsample(conv1x1(x)) exactly (a 1x1 conv acts per pixel), so it is evaluated on the
        # low-resolution map (4x fewer FLOPs and bytes) and upsampled inside the last conv's residual add
        kw = dict(calls=calls, layout=layout)
        sc = self.conv_1x1(x, **kw)
        x2, st2 = self.first_conv_3x3(x, bn=self.bn1.prepare(x, calls, layout, in_stats), upsample=True, want_stats=True, **kw)
        return self.last_conv_3x3(x2, bn=self.bn2.prepare(x2, calls, layout, st2), residual=sc, residual_up=True,
                                  want_stats=out_stats, **kw)


class DBlock(torch.nn.Module):
    """D and 3-D block (dgmr/common.py:158-238)."""

    def __init__(self, input_channels: int = 12, output_channels: int = 12, conv_type: str = "standard",
                 first_relu: bool = True, keep_same_output: bool = False):
        super().__init__()
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.first_relu = first_relu
        self.keep_same_output = keep_same_output
        self.conv_type = conv_type
        conv2d = get_conv_layer(conv_type)
        self._pd = 2 if conv_type == "3d" else 1
        self.conv_1x1 = conv2d(input_channels, output_channels, 1)
        self.first_conv_3x3 = conv2d(input_channels, output_channels, 3)
        self.last_conv_3x3 = conv2d(output_channels, output_channels, 3)
        self.relu = torch.nn.ReLU()

    def forward(self, x: torch.Tensor, calls: int = 1, layout=None) -> torch.Tensor:
        """`calls` > 1: x is a batch of `calls` consecutive calls of this block (one spectral-norm sigma each); `layout`:
        ops.CallLayout when the groups are not in call order."""
        kw = dict(calls=calls, layout=layout)
        if self.input_channels != self.output_channels:
            # shortcut: pool(conv1x1(x)) == conv1x1(pool(x)) (a 1x1 conv acts per pixel, the bias is a constant, pooling is linear),
            # so the 1x1 conv runs on the pooled map: 4x (3-D: 8x) fewer rows, and the pooling pass reads the narrow input
            # instead of the wide output (the reference: dgmr/common.py:222-226; differs by fp32 reassociation only)
            x1 = self.conv_1x1(x if self.keep_same_output else ops.avg_pool_add(x, None, self._pd), **kw)
        else:
            x1 = x
        h = self.first_conv_3x3(x, pre_relu=self.first_relu, **kw)
        if self.keep_same_output:
            return self.last_conv_3x3(h, pre_relu=True, residual=x1, **kw)
        # last conv + pooling + shortcut in one operator (bf16 modes: "3x3 conv, then 2x2 average" as a 4x4 stride-2 pass, ops.ConvFn)
        return self.last_conv_3x3(h, pre_relu=True, residual=x1, pool_out=True, **kw)


class LBlock(torch.nn.Module):
    """Residual block for the latent stack (dgmr/common.py:241-300); plain convs, no spectral norm."""

    def __init__(self, input_channels: int = 12, output_channels: int = 12, kernel_size: int = 3, conv_type: str = "standard"):
        super().__init__()
        self.input_channels = input_channels
        self.output_channels = output_channels
        if conv_type != "standard":
            get_conv_layer(conv_type)  # raises ValueError on unknown names like the reference
        self.conv_1x1 = Conv(input_channels, output_channels - input_channels, 1)
        self.first_conv_3x3 = Conv(input_channels, output_channels, kernel_size)
        self.relu = torch.nn.ReLU()
        self.last_conv_3x3 = Conv(output_channels, output_channels, kernel_size)

    def forward(self, x) -> torch.Tensor:
        if self.input_channels < self.output_channels:
            sc = ops.cat_channels([x, self.conv_1x1(x)])
        else:
            sc = x
        x2 = self.first_conv_3x3(x, pre_relu=True)
        return self.last_conv_3x3(x2, pre_relu=True, residual=sc)


class ContextConditioningStack(torch.nn.Module, PyTorchModelHubMixin):
    """Context conditioning stack (dgmr/common.py:303-424)."""

    def __init__(self, input_channels: int = 1, output_channels: int = 768, num_context_steps: int = 4,
                 conv_type: str = "standard"):
        super().__init__()
        conv2d = get_conv_layer(conv_type)
        self.space2depth = torch.nn.PixelUnshuffle(downscale_factor=2)
        oc, ic, n = output_channels, input_channels, num_context_steps
        self.d1 = DBlock(4 * ic, ((oc // 4) * ic) // n, conv_type=conv_type)
        self.d2 = DBlock(((oc // 4) * ic) // n, ((oc // 2) * ic) // n, conv_type=conv_type)
        self.d3 = DBlock(((oc // 2) * ic) // n, (oc * ic) // n, conv_type=conv_type)
        self.d4 = DBlock((oc * ic) // n, (oc * 2 * ic) // n, conv_type=conv_type)
        self.conv1 = conv2d((oc // 4) * ic, (oc // 8) * ic, 3)
        self.conv2 = conv2d((oc // 2) * ic, (oc // 4) * ic, 3)
        self.conv3 = conv2d(oc * ic, (oc // 2) * ic, 3)
        self.conv4 = conv2d(oc * 2 * ic, oc * ic, 3)
        self.relu = torch.nn.ReLU()

    def forward(self, x: torch.Tensor, draws: int = 1, reverse: bool = False):
        """-> 4 conditioning states.  `draws` > 1 (train mode): the stack as the reference's `draws` consecutive generator forwards
        would run it on the same frames - the data is identical but every forward has its own spectral-norm sigmas, so the outputs
        differ per draw: [draws * B, C, h, w], draw-major.  `reverse`: the draws are visited last-to-first (checkpoint recompute)."""
        ops.require_hip(x, "context frames")
        b, steps = x.shape[0], x.shape[1]
        # PixelUnshuffle(2) of every frame, written channels-last and frame-major in one launch; the four context steps then
        # run through d1..d4 as ONE batch of `steps` calls (each call keeps its own spectral-norm sigma)
        s = ops.frames_s2d(x, None, pool=False, frame_major=True)
        lay_blk = lay_mix = None
        if draws > 1 or reverse:
            if draws > 1:
                s = ops.repeat_batch(s, draws)  # [draw][step][sample]
            lay_blk = ops.CallLayout(draws, steps, time_major=False, reverse=reverse)
            lay_mix = ops.CallLayout(draws, 1, time_major=False, reverse=reverse)
        scales = []
        for blk in (self.d1, self.d2, self.d3, self.d4):
            s = blk(s, calls=steps * draws, layout=lay_blk)
            scales.append(s)
        return tuple(self._mixing_layer(scales[lvl], conv, steps, draws, lay_mix) for lvl, conv in
                     enumerate((self.conv1, self.conv2, self.conv3, self.conv4)))

    def _mixing_layer(self, inputs, conv_block, steps, draws: int = 1, layout=None):
        # "b t c h w -> b (c t) h w" (common.py:423) as a channel-interleaving copy, then relu(SN-conv3x3)
        stacked = ops.time_to_channels(inputs, steps, draws)
        return conv_block(stacked, act_relu=True, calls=draws, layout=layout)


class LatentConditioningStack(torch.nn.Module, PyTorchModelHubMixin):
    """Latent conditioning stack (dgmr/common.py:427-497)."""

    def __init__(self, shape: (int, int, int) = (8, 8, 8), output_channels: int = 768, use_attention: bool = True):
        super().__init__()
        self.shape = shape
        self.use_attention = use_attention
        self.distribution = normal.Normal(loc=torch.Tensor([0.0]), scale=torch.Tensor([1.0]))
        self.conv_3x3 = SNConv(shape[0], shape[0], 3)
        self.l_block1 = LBlock(input_channels=shape[0], output_channels=output_channels // 32)
        self.l_block2 = LBlock(input_channels=output_channels // 32, output_channels=output_channels // 16)
        self.l_block3 = LBlock(input_channels=output_channels // 16, output_channels=output_channels // 4)
        if self.use_attention:
            self.att_block = AttentionLayer(input_channels=output_channels // 4, output_channels=output_channels // 4)
        self.l_block4 = LBlock(input_channels=output_channels // 4, output_channels=output_channels)

    def draw(self, x: torch.Tensor) -> torch.Tensor:
        """One latent draw [1, shape[0], h, w] on x's device.  z comes from the CPU generator exactly as in the reference
        (common.py:481-483): fixed seeds give the same latent on both sides.  Only 8*h*w floats cross PCIe."""
        z = self.distribution.sample(self.shape)
        return ops.upload(torch.permute(z, (3, 0, 1, 2)), x.device, x.dtype)  # (asynchronous; same values as `.type_as(x)`)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_latent(self.draw(x))

    def forward_latent(self, z: torch.Tensor, reverse: bool = False) -> torch.Tensor:
        """The stack applied to given draws z of shape [draws, shape[0], h, w]: one call of the reference's stack per draw (only
        the first conv carries per-call state, its spectral-norm sigma); `reverse`: calls made last-draw-first."""
        ops.require_hip(z, "latent draw")
        draws = z.shape[0]
        lay = ops.CallLayout(draws, 1, time_major=False, reverse=reverse) if (draws > 1 or reverse) else None
        z = self.conv_3x3(z, calls=draws, layout=lay)
        z = self.l_block1(z)
        z = self.l_block2(z)
        z = self.l_block3(z)
        if self.use_attention:
            z = self.att_block(z)
        z = self.l_block4(z)
        return z
