"""MI355X-native `diff_surfel_rasterization`: the operator boundary of Vidu4D Stage-3.

Public surface = the reference package's
(/root/reference/gs/submodules/diff-surfel-rasterization/diff_surfel_rasterization/__init__.py):

    GaussianRasterizationSettings   NamedTuple, 12 fields in the reference order (:158-170)
    GaussianRasterizer              nn.Module: __init__(raster_settings), markVisible(positions),
                                    forward(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                            scales=None, rotations=None, cov3D_precomp=None)
                                    -> (color (3,H,W), radii (N,) int32, allmap (8,H,W))   (:172-222)
    rasterize_gaussians             functional form (:21-42)
    _RasterizeGaussians             the autograd.Function (:44-156); backward returns 9 entries
                                    (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, None)

so `from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gs/gaussian_renderer/__init__.py:14) keeps working.  The native module behind it is
`vidu4d_amd._C` (hand-written HIP through a C ABI) instead of the CUDA pybind extension.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from .. import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU copies of the call arguments, written to snapshot_{fw,bw}.dump when a debug call fails
    (same file names and format as upstream, :83-90, :133-140)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        native_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                       cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                       rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        # the caller's host-side state (hints, deferred mode, gradient outputs): remembered for the backward, which the
        # autograd engine runs on a thread of its own, outside any `with context:` of the caller
        ctx.raster_context = rc = _C.current()
        if rs.debug:
            saved = _snapshot(native_args)
            try:
                out = _C.rasterize_gaussians(*native_args, context=rc)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _C.rasterize_gaussians(*native_args, context=rc)
        num_rendered, color, others, radii, geom_buf, binning_buf, img_buf = out

        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.binning_capacity = getattr(binning_buf, "_vidu4d_capacity", max(num_rendered, 1))
        ctx.segment_split = getattr(binning_buf, "_vidu4d_split", 0)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf,
                              binning_buf, img_buf)
        ctx.mark_non_differentiable(radii)
        return color, radii, others

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf,
         img_buf) = ctx.saved_tensors
        native_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_depth, sh,
                       rs.sh_degree, rs.campos, geom_buf, ctx.num_rendered, binning_buf, img_buf, rs.debug)
        if rs.debug:
            saved = _snapshot(native_args)
            try:
                grads = _C.rasterize_gaussians_backward(*native_args, binning_capacity=ctx.binning_capacity,
                                                        segment_split=ctx.segment_split, context=ctx.raster_context)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _C.rasterize_gaussians_backward(*native_args, binning_capacity=ctx.binning_capacity,
                                                    segment_split=ctx.segment_split, context=ctx.raster_context)
        (g_means2D, g_colors_precomp, g_opacities, g_means3D, g_cov3Ds_precomp, g_sh, g_scales,
         g_rotations) = grads
        return (g_means3D, g_means2D, g_sh, g_colors_precomp, g_opacities, g_scales, g_rotations, g_cov3Ds_precomp,
                None)


class _RasterizeFrames(torch.autograd.Function):
    """Extension: the F frames of a step in ONE launch set.  Upstream renders them one after the other
    (lab4d/nnutils/deformable_gaussian.py:1175-1228); here frame f owns rows [f] of means3D (F,N,3) / rotations (F,N,4)
    and tile grid f of a stacked grid, opacity / scales / sh are shared.  Per-frame results are those of F single calls.
    sh_rest / raw_params: the canonical parameters straight from the optimizer (see rasterize_frames)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, sh_rest, raw_params, aux_planes):
        rs0 = settings_list[0]
        cams = [(rs.viewmatrix, rs.campos, rs.tanfovx, rs.tanfovy) for rs in settings_list]
        for rs in settings_list[1:]:
            if (rs.image_height, rs.image_width, rs.sh_degree) != (rs0.image_height, rs0.image_width, rs0.sh_degree):
                raise RuntimeError("rasterize_frames: the frames of a call share the image size and the SH degree")
        empty = torch.empty(0, device=means3D.device)
        out = _C.rasterize_gaussians(rs0.bg, means3D, empty, opacities, scales, rotations, rs0.scale_modifier, empty,
                                     rs0.viewmatrix, rs0.projmatrix, rs0.tanfovx, rs0.tanfovy, rs0.image_height,
                                     rs0.image_width, sh, rs0.sh_degree, rs0.campos, rs0.prefiltered, rs0.debug,
                                     frame_cams=cams, sh_rest=sh_rest, raw_params=raw_params, aux_planes=aux_planes,
                                     context=_C.current())
        ctx.raster_context = _C.current()
        num_rendered, color, others, radii, geom_buf, binning_buf, img_buf = out
        ctx.settings, ctx.cams, ctx.num_rendered = rs0, cams, num_rendered
        ctx.binning_capacity = getattr(binning_buf, "_vidu4d_capacity", max(num_rendered, 1))
        ctx.segment_split = getattr(binning_buf, "_vidu4d_split", 0)
        ctx.split_sh, ctx.raw_params, ctx.aux_planes = sh_rest is not None, bool(raw_params), int(aux_planes)
        ctx.save_for_backward(means3D, scales, rotations, radii, sh, geom_buf, binning_buf, img_buf,
                              sh_rest if sh_rest is not None else empty)
        ctx.mark_non_differentiable(radii)
        return color, radii, others

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_others):
        rs = ctx.settings
        means3D, scales, rotations, radii, sh, geom_buf, binning_buf, img_buf, sh_rest = ctx.saved_tensors
        empty = torch.empty(0, device=means3D.device)
        g = _C.rasterize_gaussians_backward(rs.bg, means3D, radii, empty, scales, rotations, rs.scale_modifier, empty,
                                            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_others,
                                            sh, rs.sh_degree, rs.campos, geom_buf, ctx.num_rendered, binning_buf, img_buf,
                                            rs.debug, binning_capacity=ctx.binning_capacity,
                                            segment_split=ctx.segment_split, frame_cams=ctx.cams,
                                            sh_rest=sh_rest if ctx.split_sh else None, raw_params=ctx.raw_params,
                                            aux_planes=ctx.aux_planes, context=ctx.raster_context)
        g_means2D, _g_colors, g_opacities, g_means3D, _g_T, g_sh, g_scales, g_rotations = g
        g_sh, g_sh_rest = g_sh if ctx.split_sh else (g_sh, None)
        return g_means3D, g_means2D, g_sh, g_opacities, g_scales, g_rotations, None, g_sh_rest, None, None


def rasterize_frames(means3D, means2D, sh, opacities, scales, rotations, settings_list, sh_rest=None, raw_params=False,
                     aux_planes=0):
    """means3D (F,N,3), means2D (F,N,3) [receives the screen-space statistic], rotations (F,N,4); sh (N,M,3), opacities
    (N,1), scales (N,2) shared; settings_list: F GaussianRasterizationSettings (same size / SH degree / background).
    -> color (3,F,H,W), radii (F,N), allmap (8,F,H,W): frame f is color[:, f], allmap[:, f].

    The canonical parameters can be handed over as the optimizer holds them (gs/scene/gaussian_model.py:47-57, :98-118):
    `sh` = `_features_dc` (N,1,3) with `sh_rest` = `_features_rest` (N,15,3) instead of their concatenation, and with
    raw_params=True `scales` = `_scaling` (log-scales), `opacities` = `_opacity` (logits); the kernels activate them and
    the backward returns the gradients w.r.t. the raw tensors -- same values as exp / sigmoid / cat in torch, without
    their launches and their backward's.

    aux_planes: bit mask of the allmap planes the caller reads (0 = all).  `AUX_ALPHA` (plane 1 only; colour + silhouette
    losses) selects the colour + alpha blend: color and allmap[1] are bit-identical to the full call's, the other planes
    are zeros, and the backward TAKES the gradients of those planes as zero."""
    return _RasterizeFrames.apply(means3D, means2D, sh, opacities, scales, rotations, tuple(settings_list), sh_rest,
                                  raw_params, aux_planes)


AUX_ALPHA = 0x02  # (VIDU4D_AUX_ALPHA)
AUX_GEOM = 0x1F   # (VIDU4D_AUX_GEOM) planes 0-4: depth, alpha, normal -- colour and these planes bit-identical to the full
                  # call's, planes 5-7 (median depth, distortion, median weight) zeros, their gradients TAKEN as zero


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points in front of the near plane (z_view > 0.2)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        def absent():
            return torch.empty(0, dtype=torch.float32, device=means3D.device)

        shs = absent() if shs is None else shs
        colors_precomp = absent() if colors_precomp is None else colors_precomp
        scales = absent() if scales is None else scales
        rotations = absent() if rotations is None else rotations
        cov3D_precomp = absent() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
