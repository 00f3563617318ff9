"""skillful_nowcasting_amd — MI355X-native DGMR training step behind the reference's module API.

Mirrors ``dgmr/__init__.py:3-6`` of openclimatefix/skillful_nowcasting: same public classes, constructor
signatures, attribute names and ``state_dict`` keys; the arithmetic runs as hand-written HIP kernels for
gfx950 in ``lib/libdgmr_hip.so`` (C ABI: ``include/dgmr_hip.h``).
"""
import os as _os

# The step runs its data-gradient chain, its weight gradients, the spectral-norm plans and (data parallel) the gradient exchange on
# separate HIP streams.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order:
# with RCCL's own streams in the process the weight-gradient stream landed on the SAME hardware queue as the main chain and the
# two serialised - the whole of the "+6 % with a process group" of round 5 (rocprofv3: every weight-gradient kernel on the main
# queue; 931.6 ms with 4 queues, 882.4 with 8, 870.8 without a process group; profiles/r06_force_dist_hw_queues.log).  So: eight
# queues for a rank of a multi-process job (one process per GPU - torchrun sets WORLD_SIZE), unless the user has set the variable.
# NOT for everybody: two processes with eight queues each on ONE GPU oversubscribe its queue slots and crawl (tests/test_gpu_ddp.py:
# 9 minutes instead of 20 s).  The runtime reads the variable when HIP initialises: import this package - or export the variable -
# before the first device call of the process.
if int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1:
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .common import ContextConditioningStack, LatentConditioningStack  # noqa: E402
from .dgmr import DGMR  # noqa: E402
from .discriminators import Discriminator, SpatialDiscriminator, TemporalDiscriminator  # noqa: E402
from .generators import Generator, Sampler  # noqa: E402
from .ops import deterministic, get_precision, set_deterministic, set_precision  # noqa: E402



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
