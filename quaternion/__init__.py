"""Top-level alias: `from quaternion import quaternion_mul, quaternion_conjugate`
(/root/reference/lab4d/utils/quat_transform.py:15-16) resolves to the MI355X-native ops."""
from vidu4d_amd.quaternion import quaternion_conjugate, quaternion_mul  # noqa: F401
