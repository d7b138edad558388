from vidu4d_amd.gs.point_utils import depth_to_normal, depths_to_points  # noqa: F401
