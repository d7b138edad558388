"""MI355X-native `quaternion` module: `quaternion_mul`, `quaternion_conjugate`.

Drop-in for /root/reference/lab4d/third_party/quaternion (quaternion.py:11-121), which
lab4d/utils/quat_transform.py:15-16 imports as `from quaternion import quaternion_mul,
quaternion_conjugate` (a top-level alias package `quaternion/` at the repo root provides that name).
Differences: importing it never invokes a compiler (upstream JIT-builds with nvcc on import,
backend.py:32-40); leading batch dimensions of any rank are accepted (upstream requires (B, D) and
its caller flattens, quat_transform.py:106-117); broadcasting is not supported, as upstream.

quaternion_mul(a (..., 3|4), b (..., 3|4)) -> (..., 4), w first; a 3-vector is a pure quaternion.
Differentiable twice (the double-backward is its own HIP kernel, as upstream).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _lib


def _prep(t):
    if not t.is_cuda:
        raise RuntimeError("quaternion ops: CUDA/HIP tensors required (use lab4d's torch formulas on CPU)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _rows(t):
    return t.numel() // t.shape[-1] if t.numel() else 0


class _QMulBackward(Function):
    @staticmethod
    def forward(ctx, grad, a, b):
        grad, a, b = _prep(grad), _prep(a), _prep(b)
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        lib = _lib.load()
        _lib.check(lib.vidu4d_quaternion_mul_backward(_rows(a), grad.data_ptr(), a.data_ptr(), a.shape[-1],
                                                      b.data_ptr(), b.shape[-1], ga.data_ptr(), gb.data_ptr(),
                                                      _stream(a)), "quaternion_mul backward")
        ctx.save_for_backward(grad, a, b)
        return ga, gb

    @staticmethod
    def backward(ctx, gg_a, gg_b):
        grad, a, b = ctx.saved_tensors
        gg_a = torch.zeros_like(a) if gg_a is None else _prep(gg_a)
        gg_b = torch.zeros_like(b) if gg_b is None else _prep(gg_b)
        gg_out, g_a, g_b = torch.empty_like(grad), torch.empty_like(a), torch.empty_like(b)
        lib = _lib.load()
        _lib.check(lib.vidu4d_quaternion_mul_backward_backward(
            _rows(a), gg_a.data_ptr(), gg_b.data_ptr(), grad.data_ptr(), a.data_ptr(), a.shape[-1], b.data_ptr(),
            b.shape[-1], gg_out.data_ptr(), g_a.data_ptr(), g_b.data_ptr(), _stream(a)),
            "quaternion_mul backward-backward")
        return gg_out, g_a, g_b


class _QMul(Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.shape[:-1] != b.shape[:-1] or a.shape[-1] not in (3, 4) or b.shape[-1] not in (3, 4):
            raise RuntimeError(f"quaternion_mul: incompatible shapes {tuple(a.shape)} x {tuple(b.shape)}")
        a, b = _prep(a), _prep(b)
        out = torch.empty(a.shape[:-1] + (4,), dtype=torch.float32, device=a.device)
        lib = _lib.load()
        _lib.check(lib.vidu4d_quaternion_mul(_rows(a), a.data_ptr(), a.shape[-1], b.data_ptr(), b.shape[-1],
                                             out.data_ptr(), _stream(a)), "quaternion_mul")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, grad):
        a, b = ctx.saved_tensors
        return _QMulBackward.apply(grad, a, b)


class _QConj(Function):
    @staticmethod
    def forward(ctx, q):
        if q.shape[-1] != 4:
            raise RuntimeError("quaternion_conjugate: last dimension must be 4")
        q = _prep(q)
        out = torch.empty_like(q)
        lib = _lib.load()
        _lib.check(lib.vidu4d_quaternion_conjugate(_rows(q), q.data_ptr(), out.data_ptr(), _stream(q)),
                   "quaternion_conjugate")
        return out

    @staticmethod
    def backward(ctx, grad):
        return _QConj.apply(grad)


quaternion_mul = _QMul.apply
quaternion_conjugate = _QConj.apply
