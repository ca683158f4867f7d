"""Drop-in import path: `import extensions.diff_gaussian_rasterization as dgr` resolves to the
MI355X-native implementation (see gaussiancity_amd/rasterizer.py)."""
from gaussiancity_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    GaussianRasterizerWrapper,
    RasterizeGaussiansFunction,
)
