"""Stage-3 loss terms from the rasterizer's planes in five launches (csrc/loss.hip) instead of ~80: the same arithmetic
as stage3.compute_losses (which tests/golden/refpy_losses.npz pins against lab4d/engine/model.py:586-693, :803-842,
:895-1012) applied to the per-frame (3,H,W) colour and (8,H,W) auxiliary planes, with the learnable-background
composite of DeformableGaussian.render_view (deformable_gaussian.py:1216-1218) in front and -- once the
normal-consistency regulariser is on (step > 8000) -- the depth / normal post-processing of
gs.gaussian_renderer.render (gs/gaussian_renderer/__init__.py:118-151) inside the same kernels."""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function

from .. import _lib


_unit_cache: dict = {}


def unit_gradient(device):
    """A cached device scalar 1.0 to start the backward from (`total.backward(gradient=unit_gradient(dev))`): no
    ones_like launch per step, and _Stage3Loss.backward recognises it and reads a cached (0, 0, 0, 0, 1) vector."""
    key = str(device)
    if key not in _unit_cache:
        _unit_cache[key] = (torch.ones((), dtype=torch.float32, device=device),
                            torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0], dtype=torch.float32, device=device))
    return _unit_cache[key][0]


class _Stage3Loss(Function):
    @staticmethod
    def forward(ctx, cfg, targets, bkgd, geometry, *planes):
        """geometry (normal term only): per frame (rays_d (H*W,3), rays_o (3,), view3x3 (3,3)) -- the camera's pixel
        rays and `world_view_transform[:3, :3].T` -- or (None, None, view3x3) when the caller supplies the surf_normal
        planes: they are then the last M entries of `planes` and receive their gradient."""
        n_sn = 0
        if geometry is not None and cfg["normal_wt"] != 0.0:
            n_sn = sum(1 for g in geometry if g[1] is None)
        sn_in = planes[len(planes) - n_sn:] if n_sn else ()
        planes = planes[:len(planes) - n_sn] if n_sn else planes
        stacked = len(planes) == 2 and planes[0].dim() == 4  # (3,M,H,W), (8,M,H,W) from one stacked rasterizer call
        if stacked:
            M = planes[0].shape[1]
            dev = planes[0].device
        else:
            M = len(planes) // 2
            dev = planes[0].device
        if not planes[0].is_cuda:
            raise RuntimeError("stage3_loss: HIP tensors required")
        if M > _lib.LOSS_MAX_FRAMES:
            raise RuntimeError(f"stage3_loss: at most {_lib.LOSS_MAX_FRAMES} frames per call")
        H, W = planes[0].shape[-2:]
        keep = [t.detach().float().contiguous() for t in planes]
        if stacked:  # per-frame base pointers into the stacked tensors, planes M*H*W apart
            colors = [keep[0][:, m] for m in range(M)]
            allmaps = [keep[1][:, m] for m in range(M)]
        else:
            colors, allmaps = keep[:M], keep[M:]
        tg = {k: targets[k].detach().float().contiguous() for k in ("rgb", "mask", "vis2d")}
        if tuple(tg["rgb"].shape) != (M, H, W, 3) or tg["mask"].numel() != M * H * W or tg["vis2d"].numel() != M * H * W:
            raise RuntimeError("stage3_loss: targets must be rgb (M,H,W,3), mask / vis2d (M,H,W,1) of the rendered size")
        det = targets.get("is_detected")
        det = None if det is None else det.detach().float().contiguous()
        bg = None if bkgd is None else bkgd.detach().float().contiguous()
        a = _lib.Stage3LossArgs()
        a.M, a.H, a.W = M, H, W
        a.plane_stride = M * H * W if stacked else 0
        for m in range(M):
            a.color[m], a.allmap[m] = colors[m].data_ptr(), allmaps[m].data_ptr()
        a.bkgd = None if bg is None else bg.data_ptr()
        a.rgb, a.mask, a.vis2d = tg["rgb"].data_ptr(), tg["mask"].data_ptr(), tg["vis2d"].data_ptr()
        a.det = None if det is None else det.data_ptr()
        a.lambda_dssim, a.rgb_wt, a.mask_wt, a.dist_wt = cfg["lambda_dssim"], cfg["rgb_wt"], cfg["mask_wt"], cfg["dist_wt"]
        a.normal_wt, a.depth_ratio = float(cfg.get("normal_wt", 0.0)), float(cfg.get("depth_ratio", 0.0))
        geo_keep = []
        ctx.n_sn = n_sn
        if a.normal_wt != 0.0:
            if geometry is None or len(geometry) != M or n_sn not in (0, M):
                raise RuntimeError("stage3_loss: the normal term needs (rays_d, rays_o, view3x3) or (surf_normal, None, view3x3) per frame")
            for m, g in enumerate(geometry):
                view = g[2].detach().float().contiguous()
                a.view3x3[m] = view.data_ptr()
                if n_sn:
                    sn = sn_in[m].detach().float().contiguous()
                    if tuple(sn.shape) != (3, H, W):
                        raise RuntimeError("stage3_loss: surf_normal planes must be (3,H,W)")
                    a.surf_normal[m] = sn.data_ptr()
                    geo_keep += [view, sn]
                else:
                    rd, ro = g[0].detach().float().contiguous(), g[1].detach().float().contiguous()
                    if rd.numel() != 3 * H * W:
                        raise RuntimeError("stage3_loss: rays_d must be (H*W,3) of the rendered size")
                    a.rays_d[m], a.rays_o[m] = rd.data_ptr(), ro.data_ptr()
                    geo_keep += [view, rd, ro]
            if not n_sn:
                sd = torch.empty(M * H * W, dtype=torch.float32, device=dev)
                a.surf_depth = sd.data_ptr()
                geo_keep.append(sd)
        sums = torch.empty(_lib.LOSS_SUMS_FLOATS, dtype=torch.float32, device=dev)
        partials = torch.empty(_lib.LOSS_BLOCKS * 16, dtype=torch.float32, device=dev)
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        a.sums, a.partials, a.losses = sums.data_ptr(), partials.data_ptr(), losses.data_ptr()
        _lib.check(_lib.load().vidu4d_stage3_loss_forward(a, torch.cuda.current_stream(dev).cuda_stream), "stage3 loss forward")
        ctx.args, ctx.M, ctx.has_bg, ctx.stacked = a, M, bg is not None, stacked
        ctx.keep = (keep, tg, det, bg, sums, partials, losses, geo_keep)  # everything the argument struct points to
        # five scalar outputs (views of the kernel's output vector: the four terms and their sum): indexing ONE output
        # tensor afterwards would cost a zero-filled (5,) tensor, a copy and an add per term in the backward, and summing
        # the terms in torch two launches forward and two backward
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2], losses[3], losses[4]

    @staticmethod
    def backward(ctx, g_rgb, g_mask, g_dist, g_normal, g_total):
        keep = ctx.keep[0]
        M, dev = ctx.M, keep[0].device
        unit = _unit_cache.get(str(dev))
        if (g_rgb is None and g_mask is None and g_dist is None and g_normal is None and g_total is not None and unit is not None
                and g_total.data_ptr() == unit[0].data_ptr()):
            g = unit[1]  # the usual case: total.backward(gradient=unit_gradient(dev)), nothing to assemble
        else:
            zero = None
            parts = []
            for t in (g_rgb, g_mask, g_dist, g_normal, g_total):
                if t is None:
                    zero = torch.zeros((), dtype=torch.float32, device=dev) if zero is None else zero
                    t = zero
                parts.append(t.detach().float().reshape(()))
            g = torch.stack(parts)  # (5,) device vector the kernel reads
        if ctx.stacked:
            full = [torch.empty_like(keep[0]), torch.empty_like(keep[1])]
            g_color = [full[0][:, m] for m in range(M)]
            g_allmap = [full[1][:, m] for m in range(M)]
        else:
            g_color = [torch.empty_like(keep[m]) for m in range(M)]
            g_allmap = [torch.empty_like(keep[M + m]) for m in range(M)]
        g_bg = torch.empty(3, dtype=torch.float32, device=dev) if ctx.has_bg else None
        o = _lib.Stage3LossGrads()
        for m in range(M):
            o.g_color[m], o.g_allmap[m] = g_color[m].data_ptr(), g_allmap[m].data_ptr()
        o.g_bkgd = None if g_bg is None else g_bg.data_ptr()
        g_sn = [torch.empty(3, *keep[0].shape[-2:], dtype=torch.float32, device=dev) for _ in range(ctx.n_sn)]
        for m, t in enumerate(g_sn):
            o.g_surf_normal[m] = t.data_ptr()
        _lib.check(_lib.load().vidu4d_stage3_loss_backward(ctx.args, g.data_ptr(), o, torch.cuda.current_stream(dev).cuda_stream),
                   "stage3 loss backward")
        if ctx.stacked:
            return (None, None, g_bg, None, full[0], full[1]) + tuple(g_sn)
        return (None, None, g_bg, None) + tuple(g_color) + tuple(g_allmap) + tuple(g_sn)


def camera_geometry(cam):
    """(rays_d, rays_o, view3x3) of a KCamera for the normal term: the cached pixel rays of depths_to_points
    (gs/utils/point_utils.py:9-24) and the block render() rotates the normals by (gaussian_renderer/__init__.py:123)."""
    g = cam.__dict__.get("_loss_geometry")
    if g is None:
        rays_d, rays_o = cam.pixel_rays()
        g = cam.__dict__["_loss_geometry"] = (rays_d, rays_o, cam.world_view_transform[:3, :3].T.contiguous())
    return g


def stage3_loss(colors, allmaps, bkgd, batch: dict, step: int, cfg, cameras=None, surf_normals=None, views=None,
                depth_ratio: float = 0.0) -> dict:
    """colors / allmaps: per-frame (3,H,W) / (8,H,W) rasterizer outputs (BEFORE the learnable-background composite),
    or the (3,M,H,W) / (8,M,H,W) tensors of one stacked call (diff_surfel_rasterization.rasterize_frames);
    bkgd: the (3,) learnable background or None; batch as for compute_losses.  -> {"rgb", "mask", "dist_loss",
    "normal_loss"}: the same weighted terms compute_losses returns for them, and "total": their sum (computed by the
    kernel; back-propagate through it OR through the terms).
    The normal-consistency term (step > 8000, lambda_normal != 0) needs `cameras` (one KCamera per frame: the kernels
    derive rend_normal / surf_normal from the planes as render() does) -- or `surf_normals` (per-frame (3,H,W) planes
    the caller made, differentiable) with `views` (per-frame 3x3 rotation of the rendered normals, default identity)."""
    lam_d = float(cfg.lambda_dist) if step > 8000 else 0.0
    lam_n = float(cfg.lambda_normal) if step > 8000 else 0.0
    c = dict(lambda_dssim=float(cfg.lambda_dssim), rgb_wt=float(cfg.rgb_wt), mask_wt=float(cfg.mask_wt), dist_wt=lam_d,
             normal_wt=lam_n, depth_ratio=float(depth_ratio))
    geometry, extra = None, ()
    if lam_n != 0.0:
        if surf_normals is not None:
            dev = surf_normals[0].device
            eye = torch.eye(3, device=dev)
            geometry = [(None, None, eye if views is None else views[m]) for m in range(len(surf_normals))]
            extra = tuple(surf_normals)
        elif cameras is not None:
            geometry = [camera_geometry(cam) for cam in cameras]
        else:
            raise RuntimeError("stage3_loss: the normal-consistency term is on: pass cameras= (or surf_normals=)")
    if isinstance(colors, torch.Tensor):
        rgb, mask, dist, normal, total = _Stage3Loss.apply(c, batch, bkgd, geometry, colors, allmaps, *extra)
    else:
        rgb, mask, dist, normal, total = _Stage3Loss.apply(c, batch, bkgd, geometry, *colors, *allmaps, *extra)
    return {"rgb": rgb, "mask": mask, "dist_loss": dist, "normal_loss": normal, "total": total}
