"""Drop-in for the reference's native module `grid_encoder_ext` (extensions/grid_encoder/setup.py,
bindings.cpp:35-40): `forward` and `backward` with upstream's positional signatures."""
from gaussiancity_amd.grid_encoder import ext_backward as backward  # noqa: F401
from gaussiancity_amd.grid_encoder import ext_forward as forward  # noqa: F401
