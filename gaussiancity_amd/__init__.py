"""gaussiancity_amd -- MI355X-native differentiable Gaussian rasterizer for GaussianCity.

Scope: the one hot path of hzxie/GaussianCity that BASELINE.json names -- the differentiable
Gaussian rasterizer behind `extensions/diff_gaussian_rasterization` -- written from scratch
for gfx950 (hand-written HIP kernels in csrc/, C ABI in include/gcr.h) behind the reference's
own Python API.  Nothing else of GaussianCity is rebuilt here.
"""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    GaussianRasterizerWrapper,
    RasterizeGaussiansFunction,
)
from .helpers import get_gaussian_points, get_gaussian_rasterization  # noqa: F401

__all__ = [
    "GaussianRasterizationSettings",
    "GaussianRasterizer",
    "GaussianRasterizerWrapper",
    "RasterizeGaussiansFunction",
    "get_gaussian_points",
    "get_gaussian_rasterization",
]
