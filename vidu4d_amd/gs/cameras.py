"""KCamera: the intrinsics-only camera Vidu4D builds per frame (reference: gs/scene/cameras.py:72-162).

Stage-3 hands the rasterizer surfels that are already in camera space, so the extrinsics are the
identity, `camera_center` is the origin, and only the frustum extents (left/right/top/bottom at
z = 1, from Kinv, lab4d/nnutils/deformable_gaussian.py:947-953) matter: FoV = 2 atan(extent / 2).
Attribute names and conventions (row-vector / transposed 4x4 matrices) are the reference's."""
import torch
import torch.nn as nn

from .graphics_utils import getWorld2View2


def _scalar(v) -> float:
    return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


class KCamera(nn.Module):
    """All quantities are derived on the host from six numbers and uploaded once (upstream builds the
    4x4 matrices element by element on the GPU, cameras.py:106-146: ~40 tiny launches per frame)."""

    def __init__(self, H, W, left, right, top, bottom, trans=(0.0, 0.0, 0.0), scale=1.0, data_device="cuda"):
        super().__init__()
        try:
            self.data_device = torch.device(data_device)
        except Exception as e:  # same fallback as upstream
            print(e)
            print(f"[Warning] Custom device {data_device} failed, fallback to default cuda device")
            self.data_device = torch.device("cuda")
        dev = self.data_device
        left, right, top, bottom = (_scalar(v) for v in (left, right, top, bottom))
        f32 = torch.float32
        # fp32 arithmetic as upstream's tensor code does it
        l_, r_, t_, b_ = (torch.tensor(v, dtype=f32) for v in (left, right, top, bottom))
        self.FoVx = 2.0 * torch.arctan((r_ - l_) / 2.0)  # 0-dim tensors as upstream (:86-87), kept on the host
        self.FoVy = 2.0 * torch.arctan((t_ - b_) / 2.0)
        self.image_width = int(W)
        self.image_height = int(H)
        self.zfar = 10
        self.znear = 0.01
        self.trans = trans
        self.scale = scale
        R = torch.eye(3)
        T = torch.zeros(3)
        wvt = getWorld2View2(R, T, trans, scale).transpose(0, 1)

        # off-centre frustum with the reference's sign flips (cameras.py:106-146)
        zn, zf = self.znear, self.zfar
        l, r, t, b = l_ * zn, r_ * zn, t_ * zn, b_ * zn
        b, t = -t, -b
        l, r = -r, -l
        P = torch.zeros(4, 4)
        P[0, 0] = 2.0 * zn / (r - l)
        P[1, 1] = 2.0 * zn / (t - b)
        P[0, 2] = (r + l) / (r - l)
        P[1, 2] = (t + b) / (t - b)
        P[3, 2] = 1.0
        P[2, 2] = (zn + zf) / (zf - zn)
        P[2, 3] = -2 * (zf * zn) / (zf - zn)
        full = wvt @ P.transpose(0, 1)
        # one upload: [R | T | world_view | projection | full_proj]
        pack = torch.cat([R.reshape(-1), T, wvt.reshape(-1), P.transpose(0, 1).reshape(-1), full.reshape(-1)]).to(dev)
        self.R = pack[0:9].view(3, 3)
        self.T = pack[9:12]
        self.world_view_transform = pack[12:28].view(4, 4)
        self.projection_matrix = pack[28:44].view(4, 4)
        self.full_proj_transform = pack[44:60].view(4, 4)
        self.camera_center = self.world_view_transform[3, :3]
        self._c2w_host = torch.inverse(wvt.transpose(0, 1))
        self._original_image = None
        self._rays = None

    @property
    def original_image(self):
        """(1,H,W) zeros, allocated on first use (upstream allocates it for every camera, cameras.py:90)."""
        if self._original_image is None:
            self._original_image = torch.zeros(1, self.image_height, self.image_width, device=self.data_device)
        return self._original_image

    def pixel_rays(self):
        """(rays_d (H*W,3), rays_o (3,)) of the pixel grid in world space (what depths_to_points,
        gs/utils/point_utils.py:9-24, rebuilds on every call), cached on the camera."""
        if self._rays is None:
            import math
            W, H = self.image_width, self.image_height
            fx = W / (2 * math.tan(float(self.FoVx) / 2.0))
            fy = H / (2 * math.tan(float(self.FoVy) / 2.0))
            intr = torch.tensor([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
            m = (intr.inverse().T @ self._c2w_host[:3, :3].T).to(self.data_device)
            dev = self.data_device
            gx, gy = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(),
                                    indexing="xy")
            pix = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
            self._rays = (pix @ m, self._c2w_host[:3, 3].to(dev))
        return self._rays


class MiniCam:
    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx = fovy, fovx
        self.znear, self.zfar = znear, zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.inverse(world_view_transform)[3][:3]
