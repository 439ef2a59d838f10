"""Drop-in operator package: same export list as the reference's op/__init__.py:1-3."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from .rasterize import rasterize

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "rasterize"]
