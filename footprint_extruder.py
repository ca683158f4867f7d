"""Drop-in for the reference's top-level `footprint_extruder` module
(extensions/footprint_extruder/setup.py:16-22; imported at scripts/dataset_generator.py:34)."""
from gaussiancity_amd.points import get_points_from_projection  # noqa: F401
