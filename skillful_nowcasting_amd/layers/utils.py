"""Convolution-type switch (mirror of dgmr/layers/utils.py:8-18).

Only "standard" (2-D) and "3d" are reachable on the DGMR path: the reference's "coord" option cannot be
spectrally normalised (CoordConv has no ``.weight``), so it is rejected here with the same
``ValueError`` the reference raises for unknown names.
"""
from functools import partial

from ..nn import SNConv


def get_conv_layer(conv_type: str = "standard"):
    """Return a factory ``f(in_channels, out_channels, kernel_size, eps) -> SNConv`` for the conv type."""
    if conv_type == "standard":
        return partial(SNConv, ndim=2)
    if conv_type == "3d":
        return partial(SNConv, ndim=3)
    raise ValueError(f"{conv_type} is not a recognized Conv method")
