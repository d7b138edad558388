"""Pseudo surface normals from a rendered depth map (reference: gs/utils/point_utils.py:9-37)."""
import math

import torch


def depths_to_points(view, depthmap):
    if hasattr(view, "pixel_rays"):  # KCamera: pixel-ray grid cached per camera
        rays_d, rays_o = view.pixel_rays()
        return depthmap.reshape(-1, 1) * rays_d + rays_o
    dev = depthmap.device
    c2w = (view.world_view_transform.T).inverse()
    W, H = view.image_width, view.image_height
    fx = W / (2 * math.tan(float(view.FoVx) / 2.0))
    fy = H / (2 * math.tan(float(view.FoVy) / 2.0))
    intrins = torch.tensor([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    gx, gy = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(), indexing="xy")
    pix = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays_d = pix @ intrins.inverse().T @ c2w[:3, :3].T
    return depthmap.reshape(-1, 1) * rays_d + c2w[:3, 3]


def depth_to_normal(view, depth):
    """(1,H,W) depth -> (H,W,3) unit normals from central differences; border pixels are zero."""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    out = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    out[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out
