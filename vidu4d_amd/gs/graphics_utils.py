"""Camera matrix helpers (reference: gs/utils/graphics_utils.py:39-76)."""
import math

import torch


def getWorld2View2(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """World->view 4x4 (column-vector convention) with the reference's re-centre/scale of the camera
    position (graphics_utils.py:39-52)."""
    dev = R.device
    Rt = torch.zeros(4, 4, device=dev)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = torch.inverse(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + torch.as_tensor(translate, dtype=torch.float32, device=dev)) * scale
    return torch.inverse(C2W).float()


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Symmetric-frustum OpenGL-style projection with z in [0,1] (graphics_utils.py:54-76)."""
    tx = math.tan(float(fovX) / 2)
    ty = math.tan(float(fovY) / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
