"""Warp / camera networks against the IMPORTED reference (fixtures written by
tests/golden/make_refpy_golden.py from /root/reference's own modules): a state dict saved from the
reference's SkinningWarp / CameraMLP loads strict=True, and every output equals the reference's."""
import os

import numpy as np
import pytest
import torch

from vidu4d_amd.lab4d import quat_transform as qt
from vidu4d_amd.lab4d.bob_warp import SkinningWarp, apply_qt_to_gaussian, dual_quaternion_skinning_qt
from vidu4d_amd.lab4d.nets import CameraMLP, make_frame_info

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=2e-5, atol=2e-6)


@pytest.fixture(scope="module")
def ref():
    nets = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)["v2"]
    arr = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "refpy_warp.npz")).items()}
    fi = make_frame_info(nets["offsets"])
    warp = SkinningWarp(fi)
    cam = CameraMLP(nets["rtmat"], frame_info=fi)
    return nets, arr, warp, cam


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, **tol), f"max abs err {err:.3e} (scale {b.abs().max().item():.3e})"


def test_state_dicts_load_strict(ref):
    nets, arr, warp, cam = ref
    assert sorted(warp.state_dict()) == sorted(nets["warp"])
    assert sorted(cam.state_dict()) == sorted(nets["camera_mlp"])
    warp.load_state_dict(nets["warp"], strict=True)
    cam.load_state_dict(nets["camera_mlp"], strict=True)


def test_articulation_and_camera_values(ref):
    nets, a, warp, cam = ref
    warp.load_state_dict(nets["warp"], strict=True)
    cam.load_state_dict(nets["camera_mlp"], strict=True)
    with torch.no_grad():
        fid = a["frame_id"]
        close(warp.articulation.time_embedding(fid), a["time_embed"])
        close(warp.articulation.time_embedding.get_mean_embedding(), a["time_embed_mean"])
        t_art, rest_art = warp.articulation.get_vals_and_mean(fid)
        close(t_art[0], a["t_art_r"])
        close(t_art[1], a["t_art_d"])
        close(rest_art[0], a["rest_art_r"])
        close(rest_art[1], a["rest_art_d"])
        cq, ct = cam.get_vals(fid)
        close(cq, a["cam_q"])
        close(ct, a["cam_t"])


def test_skinning_field_values(ref):
    nets, a, warp, cam = ref
    warp.load_state_dict(nets["warp"], strict=True)
    M, N = a["Gx"].shape[:2]
    xyz = a["xyz"][None].expand(M, -1, -1)
    rest = (a["rest_art_r"], a["rest_art_d"])
    t_art = (a["t_art_r"], a["t_art_d"])
    with torch.no_grad():
        sm = warp.skinning_model
        close(sm.bone_coords(xyz, rest), a["xyz_bone"][:, :, 0], rtol=1e-4, atol=1e-5)
        skin, delta = sm(xyz, rest, None, a["inst_id"])
        close(skin, a["skin"][:, :, 0], rtol=1e-4, atol=1e-4)
        close(delta, a["delta"][:, :, 0], rtol=1e-4, atol=1e-5)
        skin_t, delta_t = sm(xyz, t_art, a["frame_id"], a["inst_id"])
        close(skin_t, a["skin_t"][:, :, 0], rtol=1e-4, atol=1e-4)
        close(delta_t, a["delta_t"][:, :, 0], rtol=1e-4, atol=1e-5)


def test_forward_warp_values_and_gradients(ref):
    """SkinningWarp.forward -> apply_qt_to_gaussian -> field2cam, as DeformableGaussian.forward_warp."""
    nets, a, warp, cam = ref
    warp.load_state_dict(nets["warp"], strict=True)
    M, N = a["Gx"].shape[:2]
    xyz = a["xyz"].clone().requires_grad_(True)
    rot = a["rot"].clone().requires_grad_(True)
    samples = {"t_articulation": (a["t_art_r"], a["t_art_d"]), "rest_articulation": (a["rest_art_r"], a["rest_art_d"])}
    xin = xyz[None, :, None].expand(M, -1, -1, -1)
    (q, t), aux = warp(xin, a["frame_id"], a["inst_id"], samples_dict=samples, return_qt=True, return_aux=True)
    close(q, a["warp_q"])
    close(t, a["warp_t"], atol=1e-5)
    x1, r1 = apply_qt_to_gaussian(xin, rot[None].expand(M, -1, -1), q, t, M)
    cq = a["cam_q"][:, None].expand(-1, N, -1)
    ct = a["cam_t"][:, None].expand(-1, N, -1)
    x2, r2 = apply_qt_to_gaussian(x1, r1, cq, ct, M)
    close(x2, a["xyz_cam"], atol=1e-5)
    close(r2, a["rot_cam"], atol=1e-5)
    close(aux["skin_entropy"], a["skin_entropy"], rtol=1e-4, atol=1e-5)
    close(aux["delta_skin"], a["delta_skin"], rtol=1e-4, atol=1e-6)
    gx, gr = torch.autograd.grad((x2 * a["Gx"]).sum() + (r2 * a["Gr"]).sum(), (xyz, rot))
    close(gx, a["g_xyz"], rtol=2e-4, atol=2e-4)
    close(gr, a["g_rot"], rtol=2e-4, atol=2e-5)


def test_blend_values_and_gradients(ref):
    """The dual-quaternion blend with the skinning probabilities as a leaf (what csrc/lbs.hip differentiates),
    torch path on the CPU; the HIP kernel is compared with the same fixture in tests/test_gpu_lbs.py."""
    nets, a, warp, cam = ref
    M, N = a["Gx"].shape[:2]
    prob = a["skin_prob"].clone().requires_grad_(True)
    xyz = a["xyz"].clone().requires_grad_(True)
    rot = a["rot"].clone().requires_grad_(True)
    q, t = dual_quaternion_skinning_qt((a["se3_r"], a["se3_d"]), prob[None].expand(M, -1, -1))
    close(q, a["lbs_q"])
    close(t, a["lbs_t"], atol=1e-5)
    xin = xyz[None, :, None].expand(M, -1, -1, -1)
    x1, r1 = apply_qt_to_gaussian(xin, rot[None].expand(M, -1, -1), q, t, M)
    x2, r2 = apply_qt_to_gaussian(x1, r1, a["cam_q"][:, None].expand(-1, N, -1), a["cam_t"][:, None].expand(-1, N, -1), M)
    close(x2, a["lbs_xyz_cam"], atol=1e-5)
    close(r2, a["lbs_rot_cam"], atol=1e-5)
    close(qt.quaternion_translation_apply(q, t, xin.reshape(M, N, 3)), a["lbs_pts"][:, :, 0], atol=1e-5)
    gp, gx, gr = torch.autograd.grad((x2 * a["Gx"]).sum() + (r2 * a["Gr"]).sum(), (prob, xyz, rot))
    close(gp, a["lbs_g_prob"], rtol=2e-4, atol=2e-4)
    close(gx, a["lbs_g_xyz"], rtol=2e-4, atol=2e-4)
    close(gr, a["lbs_g_rot"], rtol=2e-4, atol=2e-5)


def test_camera_mlp_init_from_prior():
    """base_init + mlp_init reproduce the prior poses (pose.py:95-110; time.py:77-99)."""
    torch.manual_seed(0)
    n = 12
    ang = torch.linspace(0, 0.6, n)
    rt = torch.eye(4).repeat(n, 1, 1)
    rt[:, 0, 0], rt[:, 0, 2], rt[:, 2, 0], rt[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
    rt[:, :3, 3] = torch.tensor([0.0, 0.05, 0.3])
    cam = CameraMLP(rt, frame_info=make_frame_info([0, n]), D=2, W=64)
    loss = cam.mlp_init(termination_loss=2e-5, max_iters=5000)
    assert loss < 2e-5
    q, t = cam.get_vals()
    assert torch.allclose(qt.quaternion_translation_to_se3(q, t), rt, atol=0.05)


def test_frozen_table_bias_per_instance_is_the_per_step_evaluation(ref):
    """DeformableSurfels tabulates the delta-skin MLP's first-layer bias per instance code while the networks are
    frozen (the mean time code behind it is an MLP over all frames): row i == SkinningField.frame_bias(None, [i])."""
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    nets, a, warp, cam = ref
    warp.load_state_dict(nets["warp"], strict=True)
    sm = warp.skinning_model
    with torch.no_grad():
        tab = DeformableSurfels._frame_bias_per_instance(sm, "cpu")
        n_inst = sm.delta_field.inst_embedding.mapping.weight.shape[0]
        assert tab.shape[0] == n_inst and n_inst == len(nets["offsets"]) - 1
        for i in range(n_inst):
            close(tab[i:i + 1], sm.frame_bias(None, torch.tensor([i]), 1, "cpu"), rtol=1e-6, atol=1e-7)
