"""csrc/bone_tables_math.h on the host: the arithmetic of the bone-table kernels (values AND the dual-number gradients)
against torch autograd through the module's own torch chain -- axis-angle heads -> dual quaternions -> relative to the
rest pose; rest pose -> scaled object-to-bone map (nets.ArticulationFlatMLP.forward, quat_transform, SkinningField.
bone_affine).  The same header is what hipcc compiles into bone_tables.hip."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from vidu4d_amd.lab4d import quat_transform as qt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("bt") / "libbt_host.so"
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "vidu4d_amd", "csrc"),
                           os.path.join(ROOT, "tests", "support", "bone_tables_host.cpp"), "-o", str(out)])
    return ctypes.CDLL(str(out))


def _torch_chain(so3_t, trans_t, so3_r, trans_r, inv_gauss):
    t_art = qt.quaternion_translation_to_dual_quaternion(qt.axis_angle_to_quaternion(so3_t), trans_t)
    rest = qt.quaternion_translation_to_dual_quaternion(qt.axis_angle_to_quaternion(so3_r), trans_r)
    rest_m = (rest[0][None].expand_as(t_art[0]), rest[1][None].expand_as(t_art[1]))
    se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_m))
    q, t = qt.dual_quaternion_to_quaternion_translation(qt.dual_quaternion_inverse(rest))
    R = qt.quaternion_to_matrix(q)
    return se3[0], se3[1], (R * inv_gauss[:, :, None]).reshape(-1, 3), (t * inv_gauss).reshape(-1)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("M,B,seed,scale", [(2, 25, 0, 0.3), (1, 7, 1, 1.5), (5, 40, 2, 0.05), (3, 64, 3, 3.0)])
def test_values_and_gradients_match_autograd(host_lib, M, B, seed, scale):
    g = torch.Generator().manual_seed(seed)
    so3_t, so3_r = scale * torch.randn(M, B, 3, generator=g), scale * torch.randn(B, 3, generator=g)
    trans_t, trans_r = 0.1 * torch.randn(M, B, 3, generator=g), 0.1 * torch.randn(B, 3, generator=g)
    inv_gauss = torch.exp(-torch.log(torch.tensor(0.03)) + 0.3 * torch.randn(B, 3, generator=g))
    ins = [x.clone().requires_grad_() for x in (so3_t, trans_t, so3_r, trans_r, inv_gauss)]
    want = _torch_chain(*ins)
    gouts = [torch.randn(w.shape, generator=g) for w in want]
    want_g = torch.autograd.grad(want, ins, gouts)
    a = [x.numpy().astype(np.float32).copy() for x in (so3_t, trans_t, so3_r, trans_r, inv_gauss)]
    out = [np.zeros(tuple(w.shape), np.float32) for w in want]
    host_lib.bt_forward(M, B, *[_ptr(x) for x in a], *[_ptr(o) for o in out])
    for o, w in zip(out, want):
        assert np.abs(o - w.detach().numpy()).max() <= 2e-6 * max(1.0, float(w.abs().max()))
    go = [x.numpy().astype(np.float32).copy() for x in gouts]
    gi = [np.full_like(x, np.nan) for x in a]
    host_lib.bt_backward(M, B, *[_ptr(x) for x in a], *[_ptr(x) for x in go], *[_ptr(x) for x in gi])
    for name, got, w in zip(("so3_t", "trans_t", "so3_rest", "trans_rest", "inv_gauss"), gi, want_g):
        assert np.isfinite(got).all(), name
        assert np.abs(got - w.numpy()).max() <= 1e-5 * max(1.0, float(w.abs().max())), name


def test_zero_rotation_takes_the_series_branch(host_lib):
    """|axis-angle| < 1e-6 (an untrained head can emit exact zeros): k = 1/2 - angle^2 / 48, and the norm's derivative at
    zero is masked to zero as torch's is."""
    M, B = 1, 3
    so3_t = np.zeros((M, B, 3), np.float32)
    so3_t[0, 1] = [1e-8, 0, 0]
    trans_t = np.random.default_rng(0).normal(size=(M, B, 3)).astype(np.float32)
    so3_r, trans_r = np.zeros((B, 3), np.float32), np.zeros((B, 3), np.float32)
    ig = np.ones((B, 3), np.float32)
    out = [np.zeros(s, np.float32) for s in ((M, B, 4), (M, B, 4), (3 * B, 3), (3 * B,))]
    host_lib.bt_forward(M, B, *[_ptr(x) for x in (so3_t, trans_t, so3_r, trans_r, ig)], *[_ptr(o) for o in out])
    want = _torch_chain(*[torch.from_numpy(x) for x in (so3_t, trans_t, so3_r, trans_r, ig)])
    for o, w in zip(out, want):
        assert np.abs(o - w.numpy()).max() <= 1e-6
    assert np.allclose(out[2].reshape(B, 3, 3), np.eye(3)[None])
    gi = [np.full_like(x, np.nan) for x in (so3_t, trans_t, so3_r, trans_r, ig)]
    go = [np.ones_like(o) for o in out]
    host_lib.bt_backward(M, B, *[_ptr(x) for x in (so3_t, trans_t, so3_r, trans_r, ig)], *[_ptr(x) for x in go],
                         *[_ptr(x) for x in gi])
    assert all(np.isfinite(x).all() for x in gi)


def test_camera_tail_matches_autograd(host_lib):
    """normalize(raw) * normalize(base) (CameraMLP.get_vals' tail): values and both gradients, incl. a row whose raw
    quaternion is exactly zero (F.normalize's clamp: output 0, gradient through the clamp's constant branch)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    M = 9
    raw, base = torch.randn(M, 4, generator=g) * 3, torch.randn(M, 4, generator=g)
    raw[4] = 0
    a, b = raw.clone().requires_grad_(), base.clone().requires_grad_()
    want = qt.quaternion_mul(F.normalize(a, dim=-1), F.normalize(b, dim=-1))
    go = torch.randn(M, 4, generator=g)
    wa, wb = torch.autograd.grad(want, (a, b), go)
    r, s_, o = raw.numpy().copy(), base.numpy().copy(), np.zeros((M, 4), np.float32)
    host_lib.ct_forward(M, _ptr(r), _ptr(s_), _ptr(o))
    assert np.abs(o - want.detach().numpy()).max() <= 1e-6
    gr, gb = np.full((M, 4), np.nan, np.float32), np.full((M, 4), np.nan, np.float32)
    gon = go.numpy().copy()
    host_lib.ct_backward(M, _ptr(r), _ptr(s_), _ptr(gon), _ptr(gr), _ptr(gb))
    keep = [i for i in range(M) if i != 4]   # (at the zero row torch's gradient is g / eps-scaled: both are "whatever the clamp gives")
    assert np.abs(gr[keep] - wa.numpy()[keep]).max() <= 1e-5 * max(1.0, float(wa[keep].abs().max()))
    assert np.abs(gb - wb.numpy()).max() <= 1e-5 * max(1.0, float(wb.abs().max()))
    assert np.isfinite(gr).all() and np.isfinite(gb).all()
