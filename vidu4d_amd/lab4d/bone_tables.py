"""From the articulation heads to the fused warp's tables in one launch per direction (csrc/bone_tables.hip).

    bone_tables(so3_t (M,B,3), trans_t (M,B,3), so3_rest (B,3), trans_rest (B,3), inv_gauss (B,3))
        -> se3_qr (M,B,4), se3_qd (M,B,4), bone_A (3B,3), bone_c (3B,)

the same values as nets.ArticulationFlatMLP.forward (axis-angle, translation -> dual quaternion; reference
lab4d/nnutils/pose.py:300-323) for the frames and the rest pose, quat_transform.dual_quaternion_mul(t, inverse(rest))
(lab4d/nnutils/warping.py:415-425) and bob_warp.SkinningField.bone_affine(rest) (lab4d/nnutils/skinning.py:117-141) --
~75 elementwise launches forward and ~175 backward as torch operations.  Differentiable once (what AdamW on the networks
needs); tests/test_bone_tables_cpu.py pins the arithmetic against autograd on the host, tests/test_gpu_lbs.py the kernels."""
from __future__ import annotations

import torch

from .. import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32(t):
    return t.detach().float().contiguous()


class _BoneTables(torch.autograd.Function):
    @staticmethod
    def forward(ctx, so3_t, trans_t, so3_rest, trans_rest, inv_gauss):
        if not so3_t.is_cuda:
            raise RuntimeError("bone_tables: HIP tensors required (the torch chain is the CPU statement)")
        M, B = so3_t.shape[0], so3_t.shape[1]
        if so3_t.shape != (M, B, 3) or trans_t.shape != (M, B, 3) or so3_rest.shape != (B, 3) \
                or trans_rest.shape != (B, 3) or inv_gauss.shape != (B, 3):
            raise RuntimeError("bone_tables: shapes (M,B,3), (M,B,3), (B,3), (B,3), (B,3) expected")
        ins = tuple(_f32(t) for t in (so3_t, trans_t, so3_rest, trans_rest, inv_gauss))
        dev = so3_t.device
        qr, qd = torch.empty(M, B, 4, device=dev), torch.empty(M, B, 4, device=dev)
        A, c = torch.empty(3 * B, 3, device=dev), torch.empty(3 * B, device=dev)
        lib = _lib.load()
        _lib.check(lib.vidu4d_bone_tables_forward(M, B, *[t.data_ptr() for t in ins], qr.data_ptr(), qd.data_ptr(),
                                                  A.data_ptr(), c.data_ptr(), _stream(so3_t)), "bone_tables forward")
        ctx.save_for_backward(*ins)
        return qr, qd, A, c

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_qr, g_qd, g_A, g_c):
        ins = ctx.saved_tensors
        M, B = ins[0].shape[0], ins[0].shape[1]
        gs = [None if g is None else _f32(g) for g in (g_qr, g_qd, g_A, g_c)]
        if gs[2] is None and gs[3] is not None:   # (the kernel keys the table part on g_A)
            gs[2] = torch.zeros(3 * B, 3, device=ins[0].device)
        outs = [torch.empty_like(t) for t in ins]
        lib = _lib.load()
        _lib.check(lib.vidu4d_bone_tables_backward(M, B, *[t.data_ptr() for t in ins],
                                                   *[0 if g is None else g.data_ptr() for g in gs],
                                                   *[t.data_ptr() for t in outs], _stream(ins[0])), "bone_tables backward")
        return tuple(outs)


bone_tables = _BoneTables.apply


class _CameraTail(torch.autograd.Function):
    """cam_q = normalize(raw) * normalize(base): CameraMLP.get_vals' tail in one launch per direction."""

    @staticmethod
    def forward(ctx, raw, base):
        if not raw.is_cuda or raw.shape != base.shape or raw.shape[-1] != 4 or raw.dim() != 2:
            raise RuntimeError("camera_tail: two (M, 4) HIP tensors expected")
        r, b = _f32(raw), _f32(base)
        out = torch.empty_like(r)
        _lib.check(_lib.load().vidu4d_camera_tail_forward(r.shape[0], r.data_ptr(), b.data_ptr(), out.data_ptr(), _stream(raw)),
                   "camera_tail forward")
        ctx.save_for_backward(r, b)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        r, b = ctx.saved_tensors
        g = _f32(g)
        gr, gb = torch.empty_like(r), torch.empty_like(b)
        _lib.check(_lib.load().vidu4d_camera_tail_backward(r.shape[0], r.data_ptr(), b.data_ptr(), g.data_ptr(), gr.data_ptr(),
                                                           gb.data_ptr(), _stream(r)), "camera_tail backward")
        return gr, gb


camera_tail = _CameraTail.apply
