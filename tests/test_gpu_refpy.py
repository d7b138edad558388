"""The HIP kernels around the rasterizer (csrc/quaternion.hip, csrc/lbs.hip, csrc/post.hip) and the
GPU host path (KCamera, densify / prune surgery, losses, forward warp with the reference's own network
weights) against golden vectors produced by the IMPORTED reference Python
(tests/golden/make_refpy_golden.py -> tests/golden/refpy_*).  No expected value comes from vidu4d_amd."""
import os

import numpy as np
import pytest
import torch

from tests import test_refpy_host as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
close = H.close


def test_quaternion_hip_ops(gpu_device):
    from vidu4d_amd.quaternion import quaternion_conjugate, quaternion_mul
    r = H.load("refpy_quat.npz", gpu_device)
    close(quaternion_mul(r["a4"], r["b4"]), r["mul44"], what="mul44")
    close(quaternion_mul(r["a3"], r["b4"]), r["mul34"], what="mul34")
    close(quaternion_mul(r["a4"], r["b3"]), r["mul43"], what="mul43")
    close(quaternion_conjugate(r["a4"]), r["conj"], rtol=0, atol=0)
    a = r["a4"].clone().requires_grad_(True)
    b = r["b4"].clone().requires_grad_(True)
    Gv = r["G"].clone().requires_grad_(True)
    ga, gb = torch.autograd.grad((quaternion_mul(a, b) * Gv).sum(), (a, b), create_graph=True)
    close(ga, r["ga"], atol=1e-5), close(gb, r["gb"], atol=1e-5)
    gga, ggb, ggG = torch.autograd.grad((ga * r["Ha"]).sum() + (gb * r["Hb"]).sum(), (a, b, Gv))
    close(gga, r["gga"], atol=1e-5), close(ggb, r["ggb"], atol=1e-5), close(ggG, r["ggG"], atol=1e-5)


def test_quat_transform_on_gpu(gpu_device):
    """The (dual-)quaternion helpers with GPU tensors route through the HIP ops."""
    from vidu4d_amd.lab4d import quat_transform as qt
    r = H.load("refpy_quat.npz", gpu_device)
    close(qt.quaternion_apply(r["qn"], r["b3"]), r["q_apply"], atol=1e-5)
    dq1 = qt.quaternion_translation_to_dual_quaternion(r["qn"], r["t1"])
    dq2 = qt.quaternion_translation_to_dual_quaternion(r["pn"], r["t2"])
    close(dq1[1], r["dq1_d"])
    m = qt.dual_quaternion_mul(dq1, dq2)
    close(m[0], r["dq_mul_r"]), close(m[1], r["dq_mul_d"])
    close(qt.dual_quaternion_apply(dq1, r["b3"]), r["dq_apply"], atol=1e-5)
    close(qt.axis_angle_to_quaternion(r["aa"]), r["aa_quat"])
    close(qt.matrix_to_quaternion(r["q_matrix"]), r["matrix_q"])


def _warp_fixture(dev):
    return {k[3:]: torch.from_numpy(v).to(dev) for k, v in np.load(os.path.join(G, "refpy_warp.npz")).items()}


def test_lbs_hip_kernel(gpu_device):
    """csrc/lbs.hip (blend -> apply -> field2cam, forward and backward) vs the reference's
    dual_quaternion_skinning + apply_qt_to_gaussian x2 and their autograd gradients."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_apply
    a = _warp_fixture(gpu_device)
    prob = a["skin_prob"].clone().requires_grad_(True)
    xyz = a["xyz"].clone().requires_grad_(True)
    rot = a["rot"].clone().requires_grad_(True)
    ox, orot = lbs_apply(prob, (a["se3_r"], a["se3_d"]), xyz, rot, a["cam_q"], a["cam_t"])
    close(ox, a["lbs_xyz_cam"][:, :, 0], atol=1e-5)
    close(orot, a["lbs_rot_cam"], atol=1e-5)
    gp, gx, gr = torch.autograd.grad((ox * a["Gx"][:, :, 0]).sum() + (orot * a["Gr"]).sum(), (prob, xyz, rot))
    close(gp, a["lbs_g_prob"], rtol=2e-4, atol=2e-4)
    close(gx, a["lbs_g_xyz"], rtol=2e-4, atol=2e-4)
    close(gr, a["lbs_g_rot"], rtol=2e-4, atol=2e-5)


def _surfel_field(dev, a, nets):
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.nets import make_frame_info
    fi = make_frame_info(nets["offsets"])
    m = DeformableSurfels(dict(fg_motion="gs-bob"), num_frames=int(nets["offsets"][-1]), device=dev,
                          data_info={"frame_info": fi, "rtmat": nets["rtmat"].clone()})
    m.warp.load_state_dict(nets["warp"], strict=True)
    m.camera_mlp.load_state_dict(nets["camera_mlp"], strict=True)
    m._xyz = torch.nn.Parameter(a["xyz"].clone())
    m._rotation = torch.nn.Parameter(a["rot"].clone())
    return m


def test_forward_warp_with_reference_weights(gpu_device):
    """DeformableSurfels.forward_warp on the GPU (HIP quaternion ops underneath) with the reference's own
    warp / camera state dicts == DeformableGaussian.forward_warp of the imported reference."""
    dev = gpu_device
    a = _warp_fixture(dev)
    nets = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)["v2"]
    m = _surfel_field(dev, a, nets)
    M, N = a["Gx"].shape[:2]
    xyz = m._xyz[None, :, None].expand(M, -1, -1, -1)
    rot = m._rotation[None].expand(M, -1, -1)
    xyz_cam, rot_cam, (q, t) = m.forward_warp(xyz, rot, a["frame_id"], a["inst_id"])
    close(q, a["warp_q"]), close(t, a["warp_t"], atol=1e-5)
    close(xyz_cam, a["xyz_cam"], atol=1e-5), close(rot_cam, a["rot_cam"], atol=1e-5)
    gx, gr = torch.autograd.grad((xyz_cam * a["Gx"]).sum() + (rot_cam * a["Gr"]).sum(), (m._xyz, m._rotation))
    close(gx, a["g_xyz"], rtol=5e-4, atol=5e-4), close(gr, a["g_rot"], rtol=2e-4, atol=2e-5)
    close(m._aux_dict["skin_entropy"], a["skin_entropy"], rtol=1e-4, atol=1e-5)


def test_fused_forward_warp_with_reference_weights(gpu_device):
    """The frozen-network fast path (tabulated bones + csrc/lbs.hip) on the two frames of the fixture that
    share an instance code."""
    dev = gpu_device
    a = _warp_fixture(dev)
    nets = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)["v2"]
    m = _surfel_field(dev, a, nets)
    for mod in (m.warp, m.camera_mlp):
        for p in mod.parameters():
            p.requires_grad_(False)
    fid, iid = a["frame_id"][:2], a["inst_id"][:2]
    assert m.fused_warp_ok(iid)
    m.opts["fused_rot_activation"] = False  # the reference's forward_warp returns the orientations un-normalised
    for fused_skin in (True, False):   # skin + blend + apply in one kernel / weights in torch, blend + apply in the kernel
        m.opts["fused_skin"] = fused_skin
        m._xyz.grad = m._rotation.grad = None
        xyz_cam, rot_cam = m.forward_warp_fused(fid, iid)
        close(xyz_cam, a["f2_xyz_cam"][:, :, 0], atol=1e-5)
        close(rot_cam, a["f2_rot_cam"], atol=1e-5)
        gx, gr = torch.autograd.grad((xyz_cam * a["Gx"][:2, :, 0]).sum() + (rot_cam * a["Gr"][:2]).sum(),
                                     (m._xyz, m._rotation))
        close(gx, a["f2_g_xyz"], rtol=5e-4, atol=5e-4)
        close(gr, a["f2_g_rot"], rtol=2e-4, atol=2e-5)
    # default: the renderer's rotation activation (F.normalize, gaussian_model.py:57) is applied inside the kernel
    m.opts["fused_skin"], m.opts["fused_rot_activation"] = True, True
    _, rot_unit = m.forward_warp_fused(fid, iid)
    close(rot_unit, torch.nn.functional.normalize(a["f2_rot_cam"], dim=-1), atol=1e-5)


def test_lbs_skin_kernel_without_delta_field(gpu_device):
    """rawT = NULL (bob-nosoft): Gaussian-bone distances only, against the torch softmax + reference-pinned blend."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_apply, lbs_skin_apply
    a = _warp_fixture(gpu_device)
    N, B = a["xyz"].shape[0], a["se3_r"].shape[1]
    g = torch.Generator().manual_seed(5)
    xbT = (torch.randn(3 * B, N, generator=g) * 1.5).to(gpu_device).requires_grad_(True)
    xyz = a["xyz"].clone().requires_grad_(True)
    rot = a["rot"].clone().requires_grad_(True)
    se3 = (a["se3_r"], a["se3_d"])
    ox, orot = lbs_skin_apply(xbT, None, se3, xyz, rot, a["cam_q"], a["cam_t"])
    logits = -(xbT.view(B, 3, N) ** 2).sum(1).t()
    rx, rrot = lbs_apply(logits.softmax(-1), se3, xyz, rot, a["cam_q"], a["cam_t"])
    close(ox, rx, atol=1e-5), close(orot, rrot, atol=1e-5)
    Gx, Gr = a["Gx"][:, :, 0], a["Gr"]
    g1 = torch.autograd.grad((ox * Gx).sum() + (orot * Gr).sum(), (xbT, xyz, rot))
    g2 = torch.autograd.grad((rx * Gx).sum() + (rrot * Gr).sum(), (xbT, xyz, rot))
    for u, v in zip(g1, g2):
        close(u, v, rtol=2e-4, atol=2e-4 * float(v.abs().max()))


def test_kcamera_on_gpu(gpu_device):
    H.check_kcamera(gpu_device)


@pytest.mark.parametrize("ci,ratio", H.RENDER_CASES)
@pytest.mark.parametrize("fused", [True, False])
def test_render_postprocessing(gpu_device, ci, ratio, fused, monkeypatch):
    """render()'s depth / normal post-processing: fused HIP kernels (csrc/post.hip) and the torch chain."""
    H.check_render(ci, ratio, monkeypatch, dev=gpu_device, fused_post=fused)


def test_depth_to_normal_on_gpu(gpu_device):
    H.check_depth_to_normal(gpu_device)


def test_densify_prune_surgery_on_gpu(gpu_device):
    H.check_densify(gpu_device)


def test_device_side_densify_and_hip_adam_match_the_reference_model(gpu_device):
    """The same fixture through csrc/optim.hip: densify_and_prune_fused (plan / index / gather kernels) and SurfelAdam
    (one launch for all groups) against gs/scene/gaussian_model.py + torch.optim.Adam of the imported reference."""
    H.check_densify(gpu_device, fused=True)


@pytest.mark.parametrize("case", H.LOSS_CASES)
def test_stage3_losses_on_gpu(gpu_device, case):
    H.check_losses(case, gpu_device)
