"""Module-name shim: the reference's Python half does `import diff_gaussian_rasterization_ext`
(dgr/__init__.py:16).  With this file on sys.path that import binds to the gfx950 library."""
from gaussiancity_amd.ext import (  # noqa: F401
    mark_visible,
    rasterize_gaussians,
    rasterize_gaussians_backward,
)
