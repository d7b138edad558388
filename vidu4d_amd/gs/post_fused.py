"""Fused post-processing of the rasterizer's auxiliary planes (csrc/post.hip): the depth / normal maps
`render` derives from `allmap` (reference: gs/gaussian_renderer/__init__.py:118-151,
gs/utils/point_utils.py:9-37) in one HIP kernel per direction instead of ~25 / ~45 elementwise launches."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _SurfelPost(Function):
    @staticmethod
    def forward(ctx, allmap, rays_d, rays_o, view3x3, ratio):
        if not allmap.is_cuda:
            raise RuntimeError("surfel_post: HIP tensors required")
        allmap = allmap.detach().float().contiguous()
        _, H, W = allmap.shape
        dev = allmap.device
        rays_d, rays_o, view3x3 = (t.detach().float().contiguous() for t in (rays_d, rays_o, view3x3))
        rn = torch.empty(3, H, W, device=dev)
        med, expd, sd = (torch.empty(1, H, W, device=dev) for _ in range(3))
        sn = torch.empty(3, H, W, device=dev)
        lib = _lib.load()
        _lib.check(lib.vidu4d_post_forward(W, H, allmap.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(),
                                           view3x3.data_ptr(), float(ratio), rn.data_ptr(), med.data_ptr(),
                                           expd.data_ptr(), sd.data_ptr(), sn.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "surfel_post forward")
        ctx.save_for_backward(allmap, sd, rays_d, rays_o, view3x3)
        ctx.ratio = float(ratio)
        ctx.set_materialize_grads(False)
        return rn, med, expd, sd, sn

    @staticmethod
    def backward(ctx, g_rn, g_med, g_exp, g_sd, g_sn):
        allmap, sd, rays_d, rays_o, view3x3 = ctx.saved_tensors
        _, H, W = allmap.shape
        gs = [None if g is None else g.float().contiguous() for g in (g_rn, g_med, g_exp, g_sd, g_sn)]
        g_allmap = torch.empty_like(allmap)
        lib = _lib.load()
        _lib.check(lib.vidu4d_post_backward(W, H, allmap.data_ptr(), sd.data_ptr(), rays_d.data_ptr(), rays_o.data_ptr(),
                                            view3x3.data_ptr(), ctx.ratio, *[_ptr(g) for g in gs], g_allmap.data_ptr(),
                                            torch.cuda.current_stream(allmap.device).cuda_stream),
                   "surfel_post backward")
        return g_allmap, None, None, None, None


def surfel_post(allmap, rays_d, rays_o, view3x3, ratio: float):
    """allmap (8,H,W) -> (rend_normal (3,H,W), depth_median (1,H,W), depth_expected (1,H,W),
    surf_depth (1,H,W), surf_normal (3,H,W)); differentiable w.r.t. allmap."""
    return _SurfelPost.apply(allmap, rays_d, rays_o, view3x3, ratio)
