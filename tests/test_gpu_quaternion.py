"""HIP quaternion ops (C ABI) against the pure-torch formulas the reference falls back to on CPU
tensors (/root/reference/lab4d/utils/quat_transform.py:28-35, :63-81)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_mul(a, b):
    def w4(q):
        return q if q.shape[-1] == 4 else torch.cat([torch.zeros_like(q[..., :1]), q], -1)
    aw, ax, ay, az = w4(a).unbind(-1)
    bw, bx, by, bz = w4(b).unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


@pytest.mark.parametrize("Da,Db", [(4, 4), (3, 4), (4, 3), (3, 3)])
def test_quaternion_mul_first_and_second_order(Da, Db, gpu_device):
    from vidu4d_amd.quaternion import quaternion_conjugate, quaternion_mul
    dev = gpu_device
    g = torch.Generator().manual_seed(Da * 10 + Db)
    a = torch.randn(5, 777, Da, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(5, 777, Db, generator=g).to(dev).requires_grad_(True)
    y = quaternion_mul(a, b)
    yr = _ref_mul(a, b)
    assert y.shape == yr.shape and torch.allclose(y, yr, atol=1e-6)
    w = torch.randn(y.shape, generator=g).to(dev)
    ga, gb = torch.autograd.grad((y * w).sum(), (a, b), create_graph=True)
    gar, gbr = torch.autograd.grad((yr * w).sum(), (a, b), create_graph=True)
    assert torch.allclose(ga, gar, atol=1e-5) and torch.allclose(gb, gbr, atol=1e-5)
    # second order
    u, v = torch.randn(ga.shape, generator=g).to(dev), torch.randn(gb.shape, generator=g).to(dev)
    h = torch.autograd.grad((ga * u).sum() + (gb * v).sum(), (a, b))
    hr = torch.autograd.grad((gar * u).sum() + (gbr * v).sum(), (a, b))
    assert torch.allclose(h[0], hr[0], atol=1e-5) and torch.allclose(h[1], hr[1], atol=1e-5)
    q = torch.randn(33, 4, generator=g).to(dev).requires_grad_(True)
    c = quaternion_conjugate(q)
    assert torch.equal(c, q * torch.tensor([1.0, -1, -1, -1], device=dev))
    c.sum().backward()
    assert torch.equal(q.grad, torch.tensor([1.0, -1, -1, -1], device=dev).expand_as(q))
