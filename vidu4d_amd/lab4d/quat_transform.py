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
