"""Fused bob-LBS apply on the HIP path: skinning weights + per-frame bone dual quaternions + camera
-> camera-space surfel centres and orientations, one kernel forward and one backward (csrc/lbs.hip).

Equivalent to bob_warp.dual_quaternion_skinning_qt -> apply_qt_to_gaussian -> field2cam
apply_qt_to_gaussian (reference: lab4d/utils/geom_utils.py:48-92,
lab4d/nnutils/deformable_gaussian.py:1032-1046, :1425-1430), for bone / camera parameters that do not
require gradients (--gs_optim_warp=False)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib


def _c(t):
    return t.detach().float().contiguous()


class _LbsApply(Function):
    @staticmethod
    def forward(ctx, skin_prob, se3_qr, se3_qd, xyz, rot, cam_q, cam_t):
        if not xyz.is_cuda:
            raise RuntimeError("lbs_apply: HIP tensors required")
        for t, name in ((se3_qr, "se3"), (se3_qd, "se3"), (cam_q, "field2cam"), (cam_t, "field2cam")):
            if t.requires_grad:
                raise RuntimeError(f"lbs_apply: {name} requires grad; the fused path treats it as constant")
        M, B = se3_qr.shape[:2]
        N = xyz.shape[0]
        wT = _c(skin_prob).t().contiguous()  # (B,N): coalesced per-bone reads
        args = [wT, _c(se3_qr), _c(se3_qd), _c(xyz), _c(rot), _c(cam_q), _c(cam_t)]
        out_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=xyz.device)
        out_rot = torch.empty(M, N, 4, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        _lib.check(lib.vidu4d_lbs_forward(M, N, B, *[a.data_ptr() for a in args], out_xyz.data_ptr(),
                                          out_rot.data_ptr(), torch.cuda.current_stream(xyz.device).cuda_stream),
                   "lbs forward")
        ctx.save_for_backward(*args)
        ctx.dims = (M, N, B)
        return out_xyz, out_rot

    @staticmethod
    def backward(ctx, g_xyz_out, g_rot_out):
        M, N, B = ctx.dims
        args = ctx.saved_tensors
        dev = args[0].device
        g_xyz_out = torch.zeros(M, N, 3, device=dev) if g_xyz_out is None else _c(g_xyz_out)
        g_rot_out = torch.zeros(M, N, 4, device=dev) if g_rot_out is None else _c(g_rot_out)
        g_wT = torch.empty(M, B, N, dtype=torch.float32, device=dev)
        g_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=dev)
        g_rot = torch.empty(M, N, 4, dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.vidu4d_lbs_backward(M, N, B, *[a.data_ptr() for a in args], g_xyz_out.data_ptr(),
                                           g_rot_out.data_ptr(), g_wT.data_ptr(), g_xyz.data_ptr(), g_rot.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "lbs backward")
        return g_wT.sum(0).t(), None, None, g_xyz.sum(0), g_rot.sum(0), None, None


def lbs_apply(skin_prob, se3, xyz, rot, cam_q, cam_t):
    """skin_prob (N,B) softmax weights; se3 = (qr, qd) each (M,B,4); xyz (N,3); rot (N,4) raw
    orientations; cam_q (M,4), cam_t (M,3).  -> xyz_cam (M,N,3), rot_cam (M,N,4)."""
    return _LbsApply.apply(skin_prob, se3[0], se3[1], xyz, rot, cam_q, cam_t)


class _LbsSkinApply(Function):
    """xbT (3B,N) bone coordinates, rawT (B,N) raw delta-MLP output or None -> softmax skinning weights -> blend ->
    apply -> camera, for all frames, one kernel per direction (csrc/lbs.hip lbs_skin_kernel)."""

    @staticmethod
    def forward(ctx, xbT, rawT, se3_qr, se3_qd, xyz, rot, cam_q, cam_t):
        if not xyz.is_cuda:
            raise RuntimeError("lbs_skin_apply: HIP tensors required")
        for t, name in ((se3_qr, "se3"), (se3_qd, "se3"), (cam_q, "field2cam"), (cam_t, "field2cam")):
            if t.requires_grad:
                raise RuntimeError(f"lbs_skin_apply: {name} requires grad; the fused path treats it as constant")
        M, B = se3_qr.shape[:2]
        N = xyz.shape[0]
        if xbT.shape != (3 * B, N) or (rawT is not None and rawT.shape != (B, N)):
            raise RuntimeError(f"lbs_skin_apply: xbT {tuple(xbT.shape)} / rawT must be (3B, N) / (B, N) with B={B}, N={N}")
        args = [_c(xbT), None if rawT is None else _c(rawT), _c(se3_qr), _c(se3_qd), _c(xyz), _c(rot), _c(cam_q), _c(cam_t)]
        out_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=xyz.device)
        out_rot = torch.empty(M, N, 4, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        ptr = [None if a is None else a.data_ptr() for a in args]
        _lib.check(lib.vidu4d_lbs_skin_forward(M, N, B, *ptr, out_xyz.data_ptr(), out_rot.data_ptr(),
                                               torch.cuda.current_stream(xyz.device).cuda_stream), "lbs skin forward")
        ctx.has_raw = rawT is not None
        ctx.save_for_backward(*[a for a in args if a is not None])
        ctx.dims = (M, N, B)
        return out_xyz, out_rot

    @staticmethod
    def backward(ctx, g_xyz_out, g_rot_out):
        M, N, B = ctx.dims
        saved = list(ctx.saved_tensors)
        if not ctx.has_raw:
            saved.insert(1, None)
        dev = saved[0].device
        g_xyz_out = torch.zeros(M, N, 3, device=dev) if g_xyz_out is None else _c(g_xyz_out)
        g_rot_out = torch.zeros(M, N, 4, device=dev) if g_rot_out is None else _c(g_rot_out)
        g_xbT = torch.empty(3 * B, N, dtype=torch.float32, device=dev)
        g_rawT = torch.empty(B, N, dtype=torch.float32, device=dev) if ctx.has_raw else None
        g_xyz = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g_rot = torch.empty(N, 4, dtype=torch.float32, device=dev)
        lib = _lib.load()
        ptr = [None if a is None else a.data_ptr() for a in saved]
        _lib.check(lib.vidu4d_lbs_skin_backward(M, N, B, *ptr, g_xyz_out.data_ptr(), g_rot_out.data_ptr(), g_xbT.data_ptr(),
                                                None if g_rawT is None else g_rawT.data_ptr(), g_xyz.data_ptr(),
                                                g_rot.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "lbs skin backward")
        return g_xbT, g_rawT, None, None, g_xyz, g_rot, None, None


def lbs_skin_apply(xbT, rawT, se3, xyz, rot, cam_q, cam_t):
    """xbT (3B,N) Gaussian-bone coordinates; rawT (B,N) raw output of the delta-skin MLP or None; se3 = (qr, qd)
    each (M,B,4), M <= 8; xyz (N,3); rot (N,4); cam_q (M,4), cam_t (M,3).  -> xyz_cam (M,N,3), rot_cam (M,N,4)."""
    return _LbsSkinApply.apply(xbT, rawT, se3[0], se3[1], xyz, rot, cam_q, cam_t)
