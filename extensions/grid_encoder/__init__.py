"""Drop-in for the reference's `extensions.grid_encoder` package (extensions/grid_encoder/__init__.py):
`GridEncoder` (used at models/generator.py:37) and `GridEncoderFunction`, backed by libgce_hip.so."""
from gaussiancity_amd.grid_encoder import GridEncoder, GridEncoderFunction  # noqa: F401
