"""MI355X-native `simple_knn._C.distCUDA2` (reference: gs/submodules/simple-knn/spatial.cu:15-25,
imported by gs/scene/gaussian_model.py as `from simple_knn._C import distCUDA2`): mean squared
distance of every point to its 3 nearest other points, float32 (P,) on the points' device."""
from __future__ import annotations

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: a CUDA/HIP tensor is required (no CPU path)")
    if points.ndim != 2 or points.shape[1] != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    pts = points.detach().float().contiguous()
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    lib = _lib.load()
    _lib.check(lib.vidu4d_knn_mean_dist2(pts.shape[0], pts.data_ptr(), out.data_ptr(),
                                         torch.cuda.current_stream(pts.device).cuda_stream), "knn_mean_dist2")
    return out


def radius_neighbor_count(points: torch.Tensor, radius: float) -> torch.Tensor:
    """Points (the query included) strictly within `radius` of each point, int32 (P,): the count that
    open3d's `remove_radius_outlier(nb_points, radius)` compares with nb_points (a point is kept when
    count > nb_points; reference use: lab4d/engine/trainer.py:573-588)."""
    if not points.is_cuda:
        raise RuntimeError("radius_neighbor_count: a CUDA/HIP tensor is required (no CPU path)")
    pts = points.detach().float().contiguous()
    out = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    lib = _lib.load()
    _lib.check(lib.vidu4d_radius_count(pts.shape[0], pts.data_ptr(), float(radius), out.data_ptr(),
                                       torch.cuda.current_stream(pts.device).cuda_stream), "radius_count")
    return out
