"""Drop-in for the reference's `extensions.voxlib` package (extensions/voxlib/__init__.py:10-12): the three
functions of the native module `voxlib` (extensions/voxlib/bindings.cpp:32-40), backed by libgcv_hip.so.
The reference's Python calls the first two (scripts/dataset_generator.py:1385,1403); `maps_to_volume` has no
in-tree caller upstream and is provided for completeness of the module surface."""
from gaussiancity_amd.points import maps_to_volume, points_to_volume, ray_voxel_intersection_perspective  # noqa: F401
