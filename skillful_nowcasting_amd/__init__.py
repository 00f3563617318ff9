"""skillful_nowcasting_amd — MI355X-native DGMR training step behind the reference's module API.

Mirrors ``dgmr/__init__.py:3-6`` of openclimatefix/skillful_nowcasting: same public classes, constructor
signatures, attribute names and ``state_dict`` keys; the arithmetic runs as hand-written HIP kernels for
gfx950 in ``lib/libdgmr_hip.so`` (C ABI: ``include/dgmr_hip.h``).
"""
from .common import ContextConditioningStack, LatentConditioningStack
from .dgmr import DGMR
from .discriminators import Discriminator, SpatialDiscriminator, TemporalDiscriminator
from .generators import Generator, Sampler
from .ops import get_precision, set_precision

__all__ = ["DGMR", "Generator", "Sampler", "Discriminator", "SpatialDiscriminator", "TemporalDiscriminator",
           "ContextConditioningStack", "LatentConditioningStack", "set_precision", "get_precision"]
