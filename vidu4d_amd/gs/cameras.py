"""KCamera: the intrinsics-only camera Vidu4D builds per frame (reference: gs/scene/cameras.py:72-162).

Stage-3 hands the rasterizer surfels that are already in camera space, so the extrinsics are the
identity, `camera_center` is the origin, and only the frustum extents (left/right/top/bottom at
z = 1, from Kinv, lab4d/nnutils/deformable_gaussian.py:947-953) matter: FoV = 2 atan(extent / 2).
Attribute names and conventions (row-vector / transposed 4x4 matrices) are the reference's."""
import torch
import torch.nn as nn

from .graphics_utils import getWorld2View2


class KCamera(nn.Module):
    def __init__(self, H, W, left, right, top, bottom, trans=(0.0, 0.0, 0.0), scale=1.0, data_device="cuda"):
        super().__init__()
        try:
            self.data_device = torch.device(data_device)
        except Exception as e:  # same fallback as upstream
            print(e)
            print(f"[Warning] Custom device {data_device} failed, fallback to default cuda device")
            self.data_device = torch.device("cuda")
        dev = self.data_device
        self.R = torch.eye(3, device=dev)
        self.T = torch.zeros(3, device=dev)
        left, right, top, bottom = (torch.as_tensor(v, dtype=torch.float32, device=dev) for v in (left, right, top, bottom))
        self.FoVx = 2.0 * torch.arctan((right - left) / 2.0)
        self.FoVy = 2.0 * torch.arctan((top - bottom) / 2.0)
        self.image_width = int(W)
        self.image_height = int(H)
        self.zfar = 10
        self.znear = 0.01
        self.trans = trans
        self.scale = scale
        self.world_view_transform = getWorld2View2(self.R, self.T, trans, scale).transpose(0, 1).to(dev)

        # off-centre frustum with the reference's sign flips (cameras.py:106-146)
        zn, zf = self.znear, self.zfar
        l, r, t, b = left * zn, right * zn, top * zn, bottom * zn
        b, t = -t, -b
        l, r = -r, -l
        P = torch.zeros(4, 4, device=dev)
        P[0, 0] = 2.0 * zn / (r - l)
        P[1, 1] = 2.0 * zn / (t - b)
        P[0, 2] = (r + l) / (r - l)
        P[1, 2] = (t + b) / (t - b)
        P[3, 2] = 1.0
        P[2, 2] = (zn + zf) / (zf - zn)
        P[2, 3] = -2 * (zf * zn) / (zf - zn)
        self.projection_matrix = P.transpose(0, 1)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = self.world_view_transform[3, :3]
        self._original_image = None

    @property
    def original_image(self):
        """(1,H,W) zeros, allocated on first use (upstream allocates it for every camera, cameras.py:90)."""
        if self._original_image is None:
            self._original_image = torch.zeros(1, self.image_height, self.image_width, device=self.data_device)
        return self._original_image


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx = fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.inverse(world_view_transform)[3][:3]
