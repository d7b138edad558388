"""Quaternion / dual-quaternion algebra (w first) used by the bob warp
(reference: lab4d/utils/quat_transform.py: quaternion_mul :106-117, quaternion_apply :259-276,
quaternion_translation_apply :279-283, dual quaternions :341-469).

`quaternion_mul` / `quaternion_conjugate` dispatch like upstream: HIP kernels (vidu4d_amd.quaternion)
for GPU tensors, the torch formula on the CPU.  Operands broadcast over leading dimensions."""
from __future__ import annotations

from typing import Tuple

import torch

DualQuaternions = Tuple[torch.Tensor, torch.Tensor]


def _mul_torch(a, b):
    if a.shape[-1] == 3:
        a = torch.cat([torch.zeros_like(a[..., :1]), a], -1)
    if b.shape[-1] == 3:
        b = torch.cat([torch.zeros_like(b[..., :1]), b], -1)
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if a.is_cuda:
        from ..quaternion import quaternion_mul as _hip_mul
        shape = torch.broadcast_shapes(a.shape[:-1], b.shape[:-1])
        a = a.expand(shape + a.shape[-1:]).contiguous()
        b = b.expand(shape + b.shape[-1:]).contiguous()
        return _hip_mul(a, b)
    return _mul_torch(a, b)


def quaternion_conjugate(q: torch.Tensor) -> torch.Tensor:
    if q.is_cuda:
        from ..quaternion import quaternion_conjugate as _hip_conj
        return _hip_conj(q)
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def quaternion_apply(q: torch.Tensor, point: torch.Tensor) -> torch.Tensor:
    """q * (0, p) * conj(q); the 3-vector operand uses the kernels' pure-quaternion form."""
    return quaternion_mul(quaternion_mul(q, point), quaternion_conjugate(q))[..., 1:]


def quaternion_translation_apply(q, t, point):
    return quaternion_apply(q, point) + t


def quaternion_translation_to_dual_quaternion(q, t) -> DualQuaternions:
    return q, 0.5 * quaternion_mul(t, q)


def dual_quaternion_to_quaternion_translation(dq: DualQuaternions):
    q_r, q_d = dq
    return q_r, 2 * quaternion_mul(q_d, quaternion_conjugate(q_r))[..., 1:]


def dual_quaternion_mul(dq1: DualQuaternions, dq2: DualQuaternions) -> DualQuaternions:
    r1, d1 = dq1
    r2, d2 = dq2
    return quaternion_mul(r1, r2), quaternion_mul(r1, d2) + quaternion_mul(d1, r2)


def dual_quaternion_inverse(dq: DualQuaternions) -> DualQuaternions:
    """For unit dual quaternions: the quaternion conjugate of both parts (quat_transform.py:466-468)."""
    return quaternion_conjugate(dq[0]), quaternion_conjugate(dq[1])


def dual_quaternion_apply(dq: DualQuaternions, point):
    q, t = dual_quaternion_to_quaternion_translation(dq)
    return quaternion_translation_apply(q, t, point)


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angle = axis_angle.norm(dim=-1, keepdim=True)
    half = 0.5 * angle
    small = angle.abs() < 1e-6
    k = torch.where(small, 0.5 - angle * angle / 48, torch.sin(half) / torch.where(small, torch.ones_like(angle), angle))
    return torch.cat([torch.cos(half), axis_angle * k], -1)


def quaternion_translation_inverse(q, t):
    q_inv = quaternion_conjugate(q)
    return q_inv, quaternion_apply(q_inv, -t)


def quaternion_translation_mul(qt1, qt2):
    (q1, t1), (q2, t2) = qt1, qt2
    return quaternion_mul(q1, q2), quaternion_apply(q1, t2) + t1


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """(..., 4) w-first, any norm -> (..., 3, 3) rotation (quat_transform.py:221-255: the 2/|q|^2 form)."""
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    rows = (1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
            s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
            s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y))
    return torch.stack(rows, -1).view(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    """(..., 3, 3) -> (..., 4): of the four algebraically equal candidates (one per largest component)
    the best conditioned one (quat_transform.py:484-535)."""
    lead = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(lead + (9,)).unbind(-1)
    diag = torch.stack((1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22), -1)
    q_abs = torch.sqrt(diag.clamp_min(0))
    cand = torch.stack((torch.stack((q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), -1),
                        torch.stack((m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20), -1),
                        torch.stack((m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21), -1),
                        torch.stack((m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2), -1)), -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(-1)
    return torch.gather(cand, -2, best[..., None, None].expand(lead + (1, 4))).squeeze(-2)


def quaternion_translation_to_se3(q, t):
    rt = torch.cat((quaternion_to_matrix(q), t[..., None]), -1)
    bottom = torch.zeros_like(rt[..., :1, :])
    bottom[..., 0, 3] = 1
    return torch.cat((rt, bottom), -2)
