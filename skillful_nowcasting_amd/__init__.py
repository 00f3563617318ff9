"""skillful_nowcasting_amd — MI355X-native DGMR training step behind the reference's module API.

Mirrors ``dgmr/__init__.py:3-6`` of openclimatefix/skillful_nowcasting: same public classes, constructor
signatures, attribute names and ``state_dict`` keys; the arithmetic runs as hand-written HIP kernels for
gfx950 in ``lib/libdgmr_hip.so`` (C ABI: ``include/dgmr_hip.h``).
"""
from .common import ContextConditioningStack, LatentConditioningStack
from .dgmr import DGMR
from .discriminators import Discriminator, SpatialDiscriminator, TemporalDiscriminator
from .generators import Generator, Sampler
from .ops import deterministic, get_precision, set_deterministic, set_precision



def install_as(name: str = "dgmr"):
    """Register this package under another import name, submodules included: after `install_as("dgmr")` code written against the
    reference - `from dgmr import DGMR`, `from dgmr.layers.ConvGRU import ConvGRUCell`, `from dgmr.common import DBlock, GBlock`
    (tests/test_model.py:3-15 of openclimatefix/skillful_nowcasting) - imports these modules instead.  Refuses to shadow a
    different package that is already imported under that name."""
    import importlib
    import sys

    me = sys.modules[__name__]
    have = sys.modules.get(name)
    if have is not None and have is not me:
        raise ImportError(f"a different module is already imported as '{name}': {getattr(have, '__file__', have)}")
    sys.modules[name] = me
    for sub in ("common", "generators", "discriminators", "losses", "dgmr", "layers", "layers.ConvGRU", "layers.Attention",
                "layers.utils", "data"):
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{sub}")
    return me


__all__ = ["deterministic", "set_deterministic", "DGMR", "Generator", "Sampler", "Discriminator", "SpatialDiscriminator", "TemporalDiscriminator",
           "ContextConditioningStack", "LatentConditioningStack", "set_precision", "get_precision", "install_as"]
