"""Drop-in for the reference's `extensions.voxlib` package (extensions/voxlib/__init__.py:10-12):
same two entry points the reference's Python calls (scripts/dataset_generator.py:1385,1403), backed
by libgcv_hip.so.  `maps_to_volume` has no in-tree caller upstream and is
not provided."""
from gaussiancity_amd.points import points_to_volume, ray_voxel_intersection_perspective  # noqa: F401


def maps_to_volume(*args, **kwargs):
    raise NotImplementedError("voxlib.maps_to_volume has no caller in GaussianCity and is not part of this build")
