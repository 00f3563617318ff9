"""Spatial / temporal discriminators (mirror of dgmr/discriminators.py) on the HIP operators."""
import torch
from huggingface_hub import PyTorchModelHubMixin

from . import ops
from .common import DBlock
from .nn import BatchNorm1d, SNLinear1, SNScope


class Discriminator(torch.nn.Module, PyTorchModelHubMixin):
    """dgmr/discriminators.py:12-44."""

    def __init__(self, input_channels: int = 12, num_spatial_frames: int = 8, conv_type: str = "standard"):
        super().__init__()
        self.spatial_discriminator = SpatialDiscriminator(input_channels=input_channels, num_timesteps=num_spatial_frames,
                                                          conv_type=conv_type)
        self.temporal_discriminator = TemporalDiscriminator(input_channels=input_channels, conv_type=conv_type)

    def forward(self, x: torch.Tensor, calls: int = 1) -> torch.Tensor:
        """`calls` > 1: x stacks the inputs of `calls` consecutive calls of the reference's discriminator ([calls * N, T, C, H, W]:
        the generator pass scores each of its draws with a separate call, dgmr/dgmr.py:186-193).  Each call keeps its own random
        frame draw, spectral-norm sigmas and BatchNorm1d batch statistics, all advanced in call order."""
        if x.shape[0] % calls:
            raise RuntimeError(f"discriminator: batch {x.shape[0]} is not divisible into {calls} calls")
        # (ops.set_precision: the discriminator forward may run in a finer arithmetic mode than the rest of the step)
        with ops.discriminator_forward_precision(), SNScope(self, (tuple(x.shape), calls)):  # all spectral-norm iterations up front
            aux = ops.branch_stream(x.device) if x.is_cuda else None
            if aux is None:
                spatial_loss = self.spatial_discriminator(x, calls=calls)
                temporal_loss = self.temporal_discriminator(x, calls=calls)
            else:
                # the two discriminators share nothing but x: the temporal one runs on a second stream beside the spatial one (its
                # deep layers are small, latency-bound launches; autograd runs its backward there too).  The spatial discriminator
                # draws its random frames on the host in the reference's order either way.
                main = torch.cuda.current_stream(x.device)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    temporal_loss = self.temporal_discriminator(x, calls=calls)
                x.record_stream(aux)
                spatial_loss = self.spatial_discriminator(x, calls=calls)
                main.wait_stream(aux)
                temporal_loss.record_stream(main)
        return torch.cat([spatial_loss, temporal_loss], dim=1)


def _sum_heads(reps, frames):
    return ops.sum_groups(reps, frames).unsqueeze(1)  # [frames*N, 1] -> [N, 1, 1]


def _frame_layout(calls: int, frames: int):
    """Groups of a frame-major batch that stacks `calls` discriminator calls: [frame][call]; reference order: call-major."""
    return ops.CallLayout(calls, frames, time_major=True) if calls > 1 else None


class TemporalDiscriminator(torch.nn.Module, PyTorchModelHubMixin):
    """dgmr/discriminators.py:47-138."""

    def __init__(self, input_channels: int = 12, num_layers: int = 3, conv_type: str = "standard"):
        super().__init__()
        self.downsample = torch.nn.AvgPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2))
        self.space2depth = torch.nn.PixelUnshuffle(downscale_factor=2)
        internal_chn = 48
        self.d1 = DBlock(4 * input_channels, internal_chn * input_channels, conv_type="3d", first_relu=False)
        self.d2 = DBlock(internal_chn * input_channels, 2 * internal_chn * input_channels, conv_type="3d")
        self.intermediate_dblocks = torch.nn.ModuleList()
        for _ in range(num_layers):
            internal_chn *= 2
            self.intermediate_dblocks.append(
                DBlock(internal_chn * input_channels, 2 * internal_chn * input_channels, conv_type=conv_type))
        self.d_last = DBlock(2 * internal_chn * input_channels, 2 * internal_chn * input_channels, keep_same_output=True,
                             conv_type=conv_type)
        self.fc = SNLinear1(2 * internal_chn * input_channels)
        self.relu = torch.nn.ReLU()
        self.bn = BatchNorm1d(2 * internal_chn * input_channels)

    def forward(self, x: torch.Tensor, calls: int = 1) -> torch.Tensor:
        ops.require_hip(x, "discriminator frames")
        # AvgPool3d((1,2,2)) + PixelUnshuffle(2) + permute to N C T H W, written once as N T H W C
        x = ops.frames_s2d(x, None, pool=True, frame_major=False, as_3d=True)
        x = self.d1(x, calls=calls)  # the 3-D blocks run once per discriminator call: groups are already in call order
        x = self.d2(x, calls=calls)
        # the reference loops over the remaining frames (discriminators.py:119-133); here they form one frame-major batch and
        # every block / head runs once, each frame keeping its own spectral-norm sigma and BatchNorm1d batch statistics
        frames = x.size(2)
        lay = _frame_layout(calls, frames)
        groups = frames * calls
        rep = ops.frames_to_batch(x)
        for d in self.intermediate_dblocks:
            rep = d(rep, calls=groups, layout=lay)
        rep = self.d_last(rep, calls=groups, layout=lay)
        rep = ops.relu_sum_hw(rep)
        rep = self.bn(rep, groups=groups, layout=lay)
        rep = self.fc(rep, calls=groups, layout=lay)
        return _sum_heads(rep, frames)


class SpatialDiscriminator(torch.nn.Module, PyTorchModelHubMixin):
    """dgmr/discriminators.py:141-232."""

    def __init__(self, input_channels: int = 12, num_timesteps: int = 8, num_layers: int = 4, conv_type: str = "standard"):
        super().__init__()
        self.num_timesteps = num_timesteps
        self.mean_pool = torch.nn.AvgPool2d(2)
        self.space2depth = torch.nn.PixelUnshuffle(downscale_factor=2)
        internal_chn = 24
        self.d1 = DBlock(4 * input_channels, 2 * internal_chn * input_channels, first_relu=False, conv_type=conv_type)
        self.intermediate_dblocks = torch.nn.ModuleList()
        for _ in range(num_layers):
            internal_chn *= 2
            self.intermediate_dblocks.append(
                DBlock(internal_chn * input_channels, 2 * internal_chn * input_channels, conv_type=conv_type))
        self.d6 = DBlock(2 * internal_chn * input_channels, 2 * internal_chn * input_channels, keep_same_output=True,
                         conv_type=conv_type)
        self.fc = SNLinear1(2 * internal_chn * input_channels)
        self.relu = torch.nn.ReLU()
        self.bn = BatchNorm1d(2 * internal_chn * input_channels)

    def forward(self, x: torch.Tensor, calls: int = 1) -> torch.Tensor:
        ops.require_hip(x, "discriminator frames")
        # frame indices come from the CPU generator exactly as in the reference (discriminators.py:199): one draw per call
        idxs = torch.stack([torch.randint(low=0, high=x.size()[1], size=(self.num_timesteps,)) for _ in range(calls)])
        idxs_dev = ops.upload(idxs, x.device, torch.int32)  # (asynchronous: the host does not wait for the queue to drain)
        frames = self.num_timesteps
        lay = _frame_layout(calls, frames)
        groups = frames * calls
        # AvgPool2d(2) + PixelUnshuffle(2) of the drawn frames, frame-major: the reference's per-frame loop
        # (discriminators.py:201-226) as one batch of `frames` calls per block
        rep = ops.frames_s2d(x, idxs_dev, pool=True, frame_major=True, idx_group=x.shape[0] // calls)
        rep = self.d1(rep, calls=groups, layout=lay)
        for d in self.intermediate_dblocks:
            rep = d(rep, calls=groups, layout=lay)
        rep = self.d6(rep, calls=groups, layout=lay)
        rep = ops.relu_sum_hw(rep)
        rep = self.bn(rep, groups=groups, layout=lay)
        rep = self.fc(rep, calls=groups, layout=lay)
        return _sum_heads(rep, frames)
