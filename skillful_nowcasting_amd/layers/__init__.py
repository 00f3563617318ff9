"""Submodules for the layers (mirror of dgmr/layers/__init__.py)."""
from .Attention import AttentionLayer
from .ConvGRU import ConvGRU
