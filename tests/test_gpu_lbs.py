"""GPU parity of the fused bob-LBS apply (csrc/lbs.hip) against the torch composite it replaces
(bob_warp.dual_quaternion_skinning_qt -> apply_qt_to_gaussian x2; reference geom_utils.py:48-92,
deformable_gaussian.py:1032-1046, :1425-1430).  Floating point: 1e-5 relative to the output scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dev, M, N, B, seed):
    g = torch.Generator().manual_seed(seed)
    from vidu4d_amd.lab4d import quat_transform as qt
    qr = torch.nn.functional.normalize(torch.randn(M, B, 4, generator=g), dim=-1)
    tr = 0.3 * torch.randn(M, B, 3, generator=g)
    se3 = qt.quaternion_translation_to_dual_quaternion(qr, tr)
    logits = 3.0 * torch.randn(N, B, generator=g)
    xyz = 0.5 * torch.randn(N, 3, generator=g)
    rot = torch.randn(N, 4, generator=g)
    cq = torch.nn.functional.normalize(torch.randn(M, 4, generator=g), dim=-1)
    ct = torch.randn(M, 3, generator=g)
    return [t.to(dev) for t in (se3[0], se3[1], logits, xyz, rot, cq, ct)]


def _composite(se3, logits, xyz, rot, cq, ct):
    from vidu4d_amd.lab4d.bob_warp import apply_qt_to_gaussian, dual_quaternion_skinning_qt
    M, N = se3[0].shape[0], xyz.shape[0]
    prob = logits.softmax(-1)[None].expand(M, -1, -1)
    q, t = dual_quaternion_skinning_qt(se3, prob)
    x = xyz[None, :, None].expand(M, -1, -1, -1)
    r = rot[None].expand(M, -1, -1)
    x1, r1 = apply_qt_to_gaussian(x, r, q, t, M)
    x2, r2 = apply_qt_to_gaussian(x1, r1, cq[:, None].expand(-1, N, -1), ct[:, None].expand(-1, N, -1), M)
    return x2[:, :, 0], r2


@pytest.mark.parametrize("M,N,B", [(2, 5000, 25), (1, 257, 25), (3, 1000, 7), (1, 1, 64)])
def test_lbs_forward_backward_matches_composite(gpu_device, M, N, B):
    from vidu4d_amd.lab4d.lbs_fused import lbs_apply
    dev = gpu_device
    qr, qd, logits, xyz, rot, cq, ct = _inputs(dev, M, N, B, seed=M * 1000 + N)
    g = torch.Generator().manual_seed(7)
    gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)
    res = {}
    for name in ("fused", "torch"):
        l, x, r = (t.clone().requires_grad_(True) for t in (logits, xyz, rot))
        if name == "fused":
            ox, orot = lbs_apply(l.softmax(-1), (qr, qd), x, r, cq, ct)
        else:
            ox, orot = _composite((qr, qd), l, x, r, cq, ct)
        ((ox * gx).sum() + (orot * gr).sum()).backward()
        res[name] = [t.detach().cpu().numpy() for t in (ox, orot, l.grad, x.grad, r.grad)]
    for a, b, what in zip(res["fused"], res["torch"], ("xyz", "rot", "g_logits", "g_xyz", "g_rot")):
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-5 * scale, (what, np.abs(a - b).max(), scale)


def test_lbs_rejects_trainable_bones_and_bad_sizes(gpu_device):
    from vidu4d_amd.lab4d.lbs_fused import lbs_apply
    dev = gpu_device
    qr, qd, logits, xyz, rot, cq, ct = _inputs(dev, 1, 16, 25, seed=0)
    with pytest.raises(RuntimeError, match="requires grad"):
        lbs_apply(logits.softmax(-1), (qr.requires_grad_(True), qd), xyz, rot, cq, ct)
    qr2, qd2, logits2, *_ = _inputs(dev, 1, 16, 65, seed=0)
    with pytest.raises(RuntimeError):
        lbs_apply(logits2.softmax(-1), (qr2, qd2), xyz, rot, cq, ct)
    ox, orot = lbs_apply(logits[:0].softmax(-1), (qr.detach(), qd), xyz[:0], rot[:0], cq, ct)
    assert ox.shape == (1, 0, 3) and orot.shape == (1, 0, 4)


def test_render_frames_fused_equals_unfused(gpu_device):
    from tests.test_gpu_stage3 import _model
    from vidu4d_amd.lab4d.stage3 import make_intrinsics_inv
    dev = gpu_device
    H = W = 96
    out = {}
    for fused in (True, False):
        m = _model(dev, seed=3, fused_warp=fused, warp_aux=True)
        for mod in (m.warp, m.camera_mlp):
            for p in mod.parameters():
                p.requires_grad_(False)
        assert m.fused_warp_ok() == fused
        fid = torch.tensor([1, 5], device=dev)
        r = m.render_frames(fid, make_intrinsics_inv(2, H, W, device=dev), [H, H], [W, W])
        (r["rendered"].mean() + r["rend_normal"].mean() + r["mask"].mean()).backward()
        out[fused] = (r["rendered"].detach(), r["surf_depth"].detach(), m._xyz.grad.clone(), m._rotation.grad.clone(),
                      m._aux_dict["skin_entropy"].detach())
    for a, b in zip(out[True], out[False]):
        scale = max(1e-3, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-3 * scale, (float((a - b).abs().max()), scale)


def _skinning_field(dev, B, D, seed, frames=12):
    from vidu4d_amd.lab4d.bob_warp import SkinningField
    from vidu4d_amd.lab4d.nets import make_frame_info
    torch.manual_seed(seed)
    sm = SkinningField(num_coords=B, frame_info=make_frame_info([0, frames]), num_inst=1, D=D, W=64).to(dev)
    with torch.no_grad():  # untrained output layers are ~0: give every layer weights that matter
        for p in sm.delta_field.parameters():
            p.add_(0.2 * torch.randn_like(p))
    for p in sm.parameters():
        p.requires_grad_(False)
    return sm


@pytest.mark.parametrize("N,B,D", [(5000, 25, 2), (193, 25, 2), (1000, 7, 1), (64, 32, 4), (1, 25, 3)])
def test_skin_field_kernel_matches_the_torch_path(gpu_device, N, B, D):
    """csrc/skin_field.hip (bone map + delta-skin MLP, one thread per surfel) against addmm + SkinningField.delta_raw_T
    (themselves checked against the imported reference SkinningField in test_refpy_nets / test_gpu_refpy)."""
    from vidu4d_amd.lab4d import quat_transform as qt
    from vidu4d_amd.lab4d.lbs_fused import prepare_skin_field, skin_field, skin_field_supported
    dev = gpu_device
    sm = _skinning_field(dev, B, D, seed=N + B)
    assert skin_field_supported(sm)
    g = torch.Generator().manual_seed(3)
    rest = qt.quaternion_translation_to_dual_quaternion(
        torch.nn.functional.normalize(torch.randn(1, B, 4, generator=g), dim=-1).to(dev), (0.2 * torch.randn(1, B, 3, generator=g)).to(dev))
    A, c0 = sm.bone_affine(rest)
    bias = sm.frame_bias(None, None, 1, dev)
    xyz0 = (0.3 * torch.randn(N, 3, generator=g)).to(dev)
    Gx, Gr = torch.randn(3 * B, N, generator=g).to(dev), torch.randn(B, N, generator=g).to(dev)
    res = {}
    for name in ("fused", "fused, weights gathered per launch", "torch"):
        xyz = xyz0.clone().requires_grad_(True)
        if name.startswith("fused"):
            tab = prepare_skin_field(sm, A, c0)
            tab["pack"] = name == "fused"      # (default: the kernels copy the image vidu4d_skin_field_pack made)
            xbT, rawT = skin_field(xyz, bias[0], tab)
            assert ("packed_fwd" in tab) == tab["pack"]
        else:
            xbT = torch.addmm(c0[:, None], A, xyz.t())
            rawT = sm.delta_raw_T(xbT, bias[0])
        ((xbT * Gx).sum() + (rawT * Gr).sum()).backward()
        res[name] = [t.detach().cpu().numpy() for t in (xbT, rawT, xyz.grad)]
    for a, b in zip(res["fused"], res["fused, weights gathered per launch"]):
        assert np.array_equal(a, b)            # the same LDS contents either way
    for a, b, what in zip(res["fused"], res["torch"], ("xbT", "rawT", "g_xyz")):
        scale = max(1e-3, float(np.abs(b).max()))
        assert a.shape == b.shape and np.isfinite(a).all()
        bad = np.abs(a - b) > 3e-5 * scale
        if what == "g_xyz":
            # a hidden unit whose pre-activation is within rounding of 0 may take the other side of the ReLU in the two
            # implementations (different summation order): its whole gradient contribution then differs for that surfel
            assert bad.any(axis=1).sum() <= max(1, N // 1000), (what, int(bad.any(axis=1).sum()), np.abs(a - b).max(), scale)
        else:
            assert not bad.any(), (what, np.abs(a - b).max(), scale)
    # relu masks really differ between surfels and the output is not degenerate
    assert float(np.abs(res["torch"][1]).max()) > 1e-2


def test_skin_field_is_what_the_fused_warp_runs(gpu_device):
    """DeformableSurfels.forward_warp_fused: skin-field kernel on == off (library GEMMs), values and gradients."""
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    dev = gpu_device
    out = {}
    for flag in (True, False):
        torch.manual_seed(0)
        m = DeformableSurfels(dict(fg_motion="gs-bob", fused_skin_field=flag), num_frames=8, device=dev)
        with torch.no_grad():
            for p in m.warp.skinning_model.delta_field.parameters():
                p.add_(0.2 * torch.randn_like(p))
        for p in list(m.warp.parameters()) + list(m.camera_mlp.parameters()):
            p.requires_grad_(False)
        rng = np.random.default_rng(1)
        m.init_from_points(rng.normal(size=(3000, 3)).astype(np.float32) * 0.2, rng.uniform(size=(3000, 3)).astype(np.float32))
        x, r = m.forward_warp_fused(torch.tensor([1, 5], device=dev))
        G = torch.randn(r.shape, generator=torch.Generator().manual_seed(4)).to(dev)  # (|r| = 1: r * r would be constant)
        (x.sum() + (r * G).sum()).backward()
        out[flag] = (x.detach(), r.detach(), m._xyz.grad.clone(), m._rotation.grad.clone())
    for a, b in zip(out[True], out[False]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))


def test_lbs_skin_unit_rotation_equals_normalize_after(gpu_device):
    """lbs_skin_apply(unit_rot=True) == F.normalize(lbs_skin_apply(...)[1]) with its backward (the renderer's rotation
    activation, gs/scene/gaussian_model.py:57, fused into the kernel)."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_skin_apply
    dev = gpu_device
    M, N, B = 2, 3000, 25
    qr, qd, logits, xyz, rot, cq, ct = _inputs(dev, M, N, B, seed=5)
    g = torch.Generator().manual_seed(2)
    xbT0 = (0.5 * torch.randn(3 * B, N, generator=g)).to(dev)
    gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)
    res = {}
    for unit in (True, False):
        xb, x, r = (t.clone().requires_grad_(True) for t in (xbT0, xyz, 2.5 * rot))
        ox, orot = lbs_skin_apply(xb, None, (qr, qd), x, r, cq, ct, unit_rot=unit)
        if not unit:
            orot = torch.nn.functional.normalize(orot, dim=-1)
        ((ox * gx).sum() + (orot * gr).sum()).backward()
        res[unit] = [t.detach() for t in (ox, orot, xb.grad, x.grad, r.grad)]
    for a, b in zip(res[True], res[False]):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-6 * float(b.abs().max()))
    assert torch.allclose(res[True][1].norm(dim=-1), torch.ones(M, N, device=dev), atol=1e-6)


@pytest.mark.parametrize("M,N,B,with_raw", [(2, 3000, 25, True), (3, 513, 25, False), (1, 100, 40, True)])
def test_lbs_skin_with_the_bone_map_equals_precomputed_coordinates(gpu_device, M, N, B, with_raw):
    """lbs_skin_apply(None, ..., bone_map=(A, c)) -- x_bone = A xyz + c evaluated inside the kernels, A^T d x_bone folded
    into the centre's gradient -- == the same call on xbT = addmm(c, A, xyz^T) made by torch (values, and the gradients
    w.r.t. xyz incl. the path through the coordinates, rot, rawT)."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_skin_apply
    dev = gpu_device
    qr, qd, logits, xyz, rot, cq, ct = _inputs(dev, M, N, B, seed=11)
    g = torch.Generator().manual_seed(7)
    A = (1.5 * torch.randn(3 * B, 3, generator=g)).to(dev)
    c = (0.3 * torch.randn(3 * B, generator=g)).to(dev)
    raw0 = torch.randn(B, N, generator=g).to(dev) if with_raw else None
    gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)
    res = {}
    for fused in (True, False):
        x, r = xyz.clone().requires_grad_(True), rot.clone().requires_grad_(True)
        raw = None if raw0 is None else raw0.clone().requires_grad_(True)
        if fused:
            ox, orot = lbs_skin_apply(None, raw, (qr, qd), x, r, cq, ct, unit_rot=True, bone_map=(A, c))
        else:
            ox, orot = lbs_skin_apply(torch.addmm(c[:, None], A, x.t()), raw, (qr, qd), x, r, cq, ct, unit_rot=True)
        ((ox * gx).sum() + (orot * gr).sum()).backward()
        res[fused] = [t.detach() for t in (ox, orot, x.grad, r.grad)] + ([] if raw is None else [raw.grad])
    for a, b, what in zip(res[True], res[False], ("xyz_cam", "rot_cam", "g_xyz", "g_rot", "g_raw")):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max())), (what, float((a - b).abs().max()))
    with pytest.raises(RuntimeError, match="bone coordinates xbT OR the bone map"):
        lbs_skin_apply(None, None, (qr, qd), xyz, rot, cq, ct)


@pytest.mark.parametrize("M,N,B", [(2, 700, 40), (1, 300, 64), (2, 500, 33)])
def test_lbs_skin_more_than_32_bones_against_torch(gpu_device, M, N, B):
    """lbs_skin_apply with a delta-skin term and MORE than 32 bones against the torch statement of the same chain
    (softmax of -(|x_bone|^2 + 0.1 relu(raw)), blend, apply, camera): values and every gradient, g_raw of the bones
    32..63 included -- the relu mask of the backward is one bit per bone (round 3 kept it in a 32-bit word; comparing two
    instances of the kernel with each other, as the bone-map test does, could not see that)."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_skin_apply
    dev = gpu_device
    qr, qd, _, xyz, rot, cq, ct = _inputs(dev, M, N, B, seed=17 + B)
    g = torch.Generator().manual_seed(3)
    xbT0 = (0.7 * torch.randn(3 * B, N, generator=g)).to(dev)
    raw0 = (3.0 * torch.randn(B, N, generator=g)).to(dev)     # about half of the logits positive, for every bone
    gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)
    res = {}
    for fused in (True, False):
        xb, raw, x, r = (t.clone().requires_grad_(True) for t in (xbT0, raw0, xyz, rot))
        if fused:
            ox, orot = lbs_skin_apply(xb, raw, (qr, qd), x, r, cq, ct)
        else:
            logits = -((xb.view(B, 3, N) ** 2).sum(1) + 0.1 * torch.relu(raw)).t()
            ox, orot = _composite((qr, qd), logits, x, r, cq, ct)
        ((ox * gx).sum() + (orot * gr).sum()).backward()
        res[fused] = [t.detach() for t in (ox, orot, xb.grad, raw.grad, x.grad, r.grad)]
    for a, b, what in zip(res[True], res[False], ("xyz_cam", "rot_cam", "g_xbT", "g_raw", "g_xyz", "g_rot")):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max())), (what, float((a - b).abs().max()))
    if B > 32:
        assert float(res[False][3][32:].abs().max()) > 0, "the bones beyond 32 must carry a gradient for this to test anything"


@pytest.mark.parametrize("M,N,B,unit", [(2, 5000, 25, True), (1, 257, 25, False), (3, 1000, 7, True), (2, 700, 40, False)])
def test_lbs_skin_gradients_of_bones_and_cameras(gpu_device, M, N, B, unit):
    """Round 5: bones and cameras that TRAIN (--gs_optim_warp=True, the reference's default,
    lab4d/config.py:157).  lbs_skin_apply's backward also returns d/d se3_qr, d/d se3_qd (M,B,4), d/d cam_q (M,4),
    d/d cam_t (M,3) -- sums over all surfels, reduced in the kernel (csrc/lbs.hip, g_params) -- against autograd through the
    torch statement of the same chain (the reference's graph: geom_utils.py:48-92, deformable_gaussian.py:1032-1046,
    :1425-1430), next to the gradients the frozen path already had.  N = 257 / 700: a last workgroup that is mostly empty."""
    from vidu4d_amd.lab4d.lbs_fused import lbs_skin_apply
    dev = gpu_device
    qr0, qd0, _, xyz, rot, cq0, ct0 = _inputs(dev, M, N, B, seed=91 + N)
    g = torch.Generator().manual_seed(5)
    xbT0 = (0.7 * torch.randn(3 * B, N, generator=g)).to(dev)
    raw0 = (3.0 * torch.randn(B, N, generator=g)).to(dev)
    gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)
    res = {}
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (xbT0, raw0, qr0, qd0, xyz, rot, cq0, ct0)]
        xb, raw, qr, qd, x, r, cq, ct = leaves
        if fused:
            ox, orot = lbs_skin_apply(xb, raw, (qr, qd), x, r, cq, ct, unit_rot=unit)
        else:
            logits = -((xb.view(B, 3, N) ** 2).sum(1) + 0.1 * torch.relu(raw)).t()
            ox, orot = _composite((qr, qd), logits, x, r, cq, ct)
            if unit:
                orot = torch.nn.functional.normalize(orot, dim=-1)
        ((ox * gx).sum() + (orot * gr).sum()).backward()
        res[fused] = [t.detach() for t in (ox, orot)] + [t.grad.detach() for t in leaves]
    names = ("xyz_cam", "rot_cam", "g_xbT", "g_raw", "g_se3_qr", "g_se3_qd", "g_xyz", "g_rot", "g_cam_q", "g_cam_t")
    for a, b, what in zip(res[True], res[False], names):
        scale = float(b.abs().max())
        assert scale > 0 or what == "g_raw", what
        # (the sums run over up to 5000 surfels in another order than torch's: 1e-5 of the tensor's scale)
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-9, (what, float((a - b).abs().max()), scale)


def test_fused_warp_with_networks_that_train_gives_every_parameter_its_gradient(gpu_device):
    """The model's fused warp with bone / articulation / camera / skinning networks that require grad (fused_warp_ok no longer
    asks for frozen networks).  (1) The warp itself -- canonical surfels -> camera-space centres and orientations of two
    frames -- against the torch chain of rounds 1-4 (forward_warp; itself pinned against the imported reference,
    tests/test_refpy_*): values, and the gradient of EVERY parameter that reaches it -- canonical centres and orientations, the
    articulation MLP, the camera MLP, the skinning field incl. its delta MLP and Gaussian-bone scales -- to 1e-5 of each
    tensor's scale.  (2) Through the rasterizer (render_frames): the same parameters receive gradients, equal to the torch
    chain's to 1e-3 of scale (the two warps round differently in the last bits, which flips a few of the rasterizer's
    per-pixel threshold decisions: test_render_frames_fused_equals_unfused holds the frozen path to the same bar)."""
    from tests.test_gpu_stage3 import _model
    from vidu4d_amd.lab4d.stage3 import make_intrinsics_inv
    dev = gpu_device
    H = W = 96

    def make(fused):
        m = _model(dev, seed=3, fused_warp=fused)
        with torch.no_grad():   # (untrained delta / articulation output layers are ~0: give them weights that matter)
            for mod in (m.warp, m.camera_mlp):
                for p in mod.parameters():
                    p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev))
        assert m.warp_networks_train() and m.fused_warp_ok() == fused
        return m

    def grads(m):
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    def compare(got, want, tol, what):
        assert set(got) == set(want), (what, set(got) ^ set(want))
        bad = {}
        for k, b in want.items():
            scale = float(b.abs().max())
            err = float((got[k] - b).abs().max())
            if err > tol * scale + 1e-12:
                bad[k] = (err, scale)
        assert not bad, (what, bad)

    fid = torch.tensor([1, 5], device=dev)
    # ---- (1) the warp alone
    res = {}
    for fused in (True, False):
        m = make(fused)
        N = m._xyz.shape[0]
        if fused:
            x, r = m.forward_warp_fused(fid)
            m.__dict__.pop("_warp_rot_is_unit", None)
            # (the networks' forward and backward ran as captured hipGraphs: DeformableSurfels._graphed_warp_networks)
            assert m.__dict__["_net_graph"][1] is not None, "the networks were evaluated eagerly: graph capture failed"
        else:
            xyz = m._xyz[None, :, None].expand(2, -1, -1, -1)
            rot = m._rotation[None].expand(2, -1, -1)
            x, r, _ = m.forward_warp(xyz, rot, fid)
            x = x[:, :, 0]
            if m.opts.get("fused_rot_activation", True):   # (the fused kernel hands the orientations on normalised)
                r = torch.nn.functional.normalize(r, dim=-1)
        gen = torch.Generator().manual_seed(11)
        gx, gr = torch.randn(2, N, 3, generator=gen).to(dev), torch.randn(2, N, 4, generator=gen).to(dev)
        ((x * gx).sum() + (r * gr).sum()).backward()
        res[fused] = (x.detach(), r.detach(), grads(m))
    for a, b, what in zip(res[True][:2], res[False][:2], ("xyz_cam", "rot_cam")):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), what
    net = [k for k in res[False][2] if k.startswith(("warp.", "camera_mlp."))]
    assert len(net) >= 10, net
    assert all(float(res[False][2][k].abs().max()) > 0 for k in net if "inst_embedding" not in k), "a network parameter without gradient"
    # (two float32 evaluations with different summation orders -- library GEMMs against the fused stacks' in-wave sums, the
    # autograd chain against the bone tables' dual numbers: 1e-5 of scale holds for all tensors but the axis-angle head's last
    # bias, a sum of opposite-signed terms, measured at 1.0-1.3e-5; the gate is 2e-5)
    compare(res[True][2], res[False][2], 2e-5, "warp")
    # ---- (2) through the rasterizer
    out = {}
    for fused in (True, False):
        m = make(fused)
        r = m.render_frames(fid, make_intrinsics_inv(2, H, W, device=dev), [H, H], [W, W])
        gen = torch.Generator().manual_seed(12)
        wts = {k: torch.randn(r[k].shape, generator=gen).to(dev) for k in ("rendered", "mask", "rend_normal")}
        sum((r[k] * wts[k]).sum() for k in wts).backward()
        out[fused] = ({k: r[k].detach() for k in wts}, grads(m))
    for k in out[False][0]:
        assert float((out[True][0][k] - out[False][0][k]).abs().max()) <= 1e-3 * float(out[False][0][k].abs().max()), k
    # (network parameters: sums over all surfels, 1e-3; a per-surfel tensor sees a flipped pixel in ONE of its rows: 1e-2)
    compare({k: v for k, v in out[True][1].items() if k in net}, {k: v for k, v in out[False][1].items() if k in net}, 1e-3, "render, networks")
    compare({k: v for k, v in out[True][1].items() if k not in net}, {k: v for k, v in out[False][1].items() if k not in net}, 1e-2,
            "render, surfels")


@pytest.mark.parametrize("M,B,seed", [(2, 25, 0), (1, 7, 1), (8, 40, 2), (3, 130, 3)])
def test_bone_table_kernels_match_the_torch_chain(gpu_device, M, B, seed):
    """csrc/bone_tables.hip (heads -> relative bone transforms, rest pose's scaled bone map) against the torch statement
    of the same chain with autograd: values to 2e-6, the five inputs' gradients to 1e-5 of scale (the arithmetic itself is
    pinned on the host by tests/test_bone_tables_cpu.py; this is the launch: indexing, more than one workgroup, NULL parts)."""
    from tests.test_bone_tables_cpu import _torch_chain
    from vidu4d_amd.lab4d.bone_tables import bone_tables
    dev = gpu_device
    g = torch.Generator().manual_seed(seed)
    ins = [0.5 * torch.randn(M, B, 3, generator=g), 0.1 * torch.randn(M, B, 3, generator=g), 0.5 * torch.randn(B, 3, generator=g),
           0.1 * torch.randn(B, 3, generator=g), torch.exp(3.5 + 0.3 * torch.randn(B, 3, generator=g))]
    gouts = [torch.randn(s, generator=g) for s in ((M, B, 4), (M, B, 4), (3 * B, 3), (3 * B,))]
    a = [x.clone().requires_grad_() for x in ins]
    want = _torch_chain(*a)
    want_g = torch.autograd.grad(want, a, gouts)
    b = [x.to(dev).requires_grad_() for x in ins]
    got = bone_tables(*b)
    for o, w in zip(got, want):
        assert float((o.cpu() - w).abs().max()) <= 2e-6 * max(1.0, float(w.abs().max()))
    got_g = torch.autograd.grad(got, b, [x.to(dev) for x in gouts])
    for name, x, w in zip(("so3_t", "trans_t", "so3_rest", "trans_rest", "inv_gauss"), got_g, want_g):
        assert float((x.cpu() - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max())), name
    # only the bone transforms' gradients arrive (the tables unused): the rest pose still gets its share, the extents zero
    got = bone_tables(*b)
    part = torch.autograd.grad(got[:2], b, [x.to(dev) for x in gouts[:2]], allow_unused=True)
    want_part = torch.autograd.grad(_torch_chain(*a)[:2], a, gouts[:2], allow_unused=True)
    for x, w in zip(part[:4], want_part[:4]):
        assert float((x.cpu() - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max()))
    assert float(part[4].abs().max()) == 0.0


def test_weight_gradient_contraction_over_the_surfels_in_chunks(gpu_device):
    """bob_warp.contract_over_columns on the device at the bench's size: the batched GEMM over strided views of the
    feature-major arrays (no copies) gives G @ X^T to float rounding; through feature_major_linear the three gradients
    equal torch.addmm's."""
    from vidu4d_amd.lab4d.bob_warp import SPLIT_K_CHUNK, contract_over_columns, feature_major_linear
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    for O, I, N in ((64, 75, 200000), (25, 64, 200000), (64, 64, 8192 + 3), (75, 3, 50001)):
        G, X = torch.randn(O, N, generator=g).to(dev), torch.randn(I, N, generator=g).to(dev)
        want = G.double() @ X.double().t()
        got = contract_over_columns(G, X)
        # (float32 sums of up to 200 000 products, in an order of the library's choosing)
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    N = 5 * SPLIT_K_CHUNK + 9
    X0 = torch.randn(75, N, generator=g).to(dev)
    gy = torch.randn(64, N, generator=g).to(dev)
    res = {}
    for split in (True, False):
        W = torch.randn(64, 75, generator=torch.Generator().manual_seed(1)).to(dev).requires_grad_()
        b = torch.randn(64, generator=torch.Generator().manual_seed(2)).to(dev).requires_grad_()
        X = X0.clone().requires_grad_()
        Y = feature_major_linear(b, W, X, SPLIT_K_CHUNK) if split else torch.addmm(b[:, None], W, X)
        Y.backward(gy)
        res[split] = (Y.detach(), W.grad, b.grad, X.grad)
    for a, c in zip(res[True], res[False]):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())


@pytest.mark.parametrize("N", [5000, 70001])
def test_skin_field_with_weights_that_train(gpu_device, N):
    """csrc/skin_field.hip's TRAIN instances (lbs_fused.skin_field_train): bone coordinates and delta-skin MLP on the matrix
    cores with EVERY input differentiable -- canonical centres, the step's first-layer bias, the bone map (A, c) and all
    weights and biases of the MLP -- against the feature-major library GEMMs with autograd."""
    from tests.test_gpu_stage3 import _model
    from vidu4d_amd.lab4d.lbs_fused import skin_field_shape_supported, skin_field_train
    dev = gpu_device
    m = _model(dev, n=N, seed=4)
    sm = m.warp.skinning_model
    assert skin_field_shape_supported(sm)
    with torch.no_grad():
        for p in sm.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).to(dev))
    B = sm.log_gauss.shape[0]
    g = torch.Generator().manual_seed(N)
    A0, c0_0 = (3.0 * torch.randn(3 * B, 3, generator=g)).to(dev), torch.randn(3 * B, generator=g).to(dev)
    bias0 = torch.randn(sm.delta_field.W, generator=g).to(dev)
    xyz0 = m._xyz.detach().clone()
    g_xb, g_raw = torch.randn(3 * B, N, generator=g).to(dev), torch.randn(B, N, generator=g).to(dev)
    def both(g_xb, g_raw):
        res = {}
        for fused in (True, False):
            sm.zero_grad(set_to_none=True)
            xyz, A, c0, bias = (t.clone().requires_grad_() for t in (xyz0, A0, c0_0, bias0))
            if fused:
                xbT, rawT = skin_field_train(xyz, bias, A, c0, sm)
            else:
                xbT = torch.addmm(c0[:, None], A, xyz.t())
                rawT = sm.delta_raw_T(xbT, bias, split_k=0)
            torch.autograd.backward([xbT, rawT], [g_xb, g_raw])
            grads = {"xyz": xyz.grad, "A": A.grad, "c": c0.grad, "bias": bias.grad}
            grads.update({k: p.grad for k, p in sm.delta_field.named_parameters() if p.grad is not None})
            res[fused] = (xbT.detach(), rawT.detach(), {k: v.clone() for k, v in grads.items()})
        return res

    res = both(g_xb, g_raw)
    for a, b in zip(res[True][:2], res[False][:2]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    # A hidden unit whose pre-activation is within rounding of zero is "on" in one evaluation and "off" in the other (two
    # summation orders; 128 units x N surfels: seen once in 70 001) -- a kink of the function, not an error of either.  Such
    # surfels show in d/d xyz, a per-surfel quantity; they are few, and with THEIR upstream gradients zeroed (a surfel's
    # share of every weight gradient is linear in them) everything must agree.
    d = (res[True][2]["xyz"] - res[False][2]["xyz"]).abs().max(1).values
    flipped = torch.nonzero(d > 2e-5 * float(res[False][2]["xyz"].abs().max())).flatten()
    assert flipped.numel() <= max(1, N // 20000), flipped.numel()
    if flipped.numel():
        g_xb, g_raw = g_xb.clone(), g_raw.clone()
        g_xb[:, flipped] = 0
        g_raw[:, flipped] = 0
        res = both(g_xb, g_raw)
    assert set(res[True][2]) == set(res[False][2]) and len(res[False][2]) >= 9, set(res[True][2]) ^ set(res[False][2])
    for k, b in res[False][2].items():
        a = res[True][2][k]
        assert float(b.abs().max()) > 0, k
        # (sums over up to 70 001 surfels in another order; 2e-5 of the tensor's scale)
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
    # the first layer's weight: only its coordinate columns enter here
    w1g = res[True][2]["linear_1.0.weight"]
    assert float(w1g[:, 3 * B:].abs().max()) == 0.0


def test_camera_tail_kernel_matches_the_torch_ops(gpu_device):
    """csrc/bone_tables.hip camera_tail: normalize(raw) * normalize(base) and both gradients on the device (the arithmetic
    is pinned on the host by tests/test_bone_tables_cpu.py)."""
    import torch.nn.functional as F
    from vidu4d_amd.lab4d import quat_transform as qt
    from vidu4d_amd.lab4d.bone_tables import camera_tail
    dev = gpu_device
    g = torch.Generator().manual_seed(1)
    for M in (1, 2, 8, 100):
        raw, base, go = (torch.randn(M, 4, generator=g).to(dev) for _ in range(3))
        res = {}
        for fused in (True, False):
            a, b = raw.clone().requires_grad_(), base.clone().requires_grad_()
            out = camera_tail(a, b) if fused else qt.quaternion_mul(F.normalize(a, dim=-1), F.normalize(b, dim=-1))
            out.backward(go)
            res[fused] = (out.detach(), a.grad, b.grad)
        for x, y in zip(res[True], res[False]):
            assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))


def test_contractions_over_the_surfels_in_one_launch(gpu_device):
    """lbs_fused.contract_pairs (csrc/contract.hip: the skinning field's weight gradients, fp32 matrix cores, one launch per
    four contractions) against the products in float64: the shapes of the bob networks' step (75 x 4 against points stored
    (N, 4), 64 x 76, 25 x 65, 64 x 65), N not a multiple of anything, rows that are strided views."""
    from vidu4d_amd.lab4d.lbs_fused import contract_pairs
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    for N in (200_000, 175_366, 33, 4097):
        pts = torch.randn(N, 4, generator=g).to(dev)
        big = torch.randn(2, 64, N, generator=g).to(dev)
        pairs = [(torch.randn(75, N, generator=g).to(dev), pts.t()),
                 (big[0], torch.randn(76, N, generator=g).to(dev)),
                 (torch.randn(25, N, generator=g).to(dev), torch.randn(66, N, generator=g).to(dev)[:65]),
                 (big[1], torch.randn(65, N, generator=g).to(dev)),
                 (torch.randn(96, N, generator=g).to(dev), torch.randn(1, N, generator=g).to(dev))]
        outs = contract_pairs(pairs)
        for (l, r), o in zip(pairs, outs):
            ref = l.double() @ r.double().t()
            err = float((o.double() - ref).abs().max())
            assert o.shape == ref.shape and err <= 2e-6 * float(ref.abs().max()) * max(1.0, (N / 1e4) ** 0.5), (N, tuple(o.shape), err)


def test_strided_copies_in_one_launch(gpu_device):
    """lbs_fused.copy_strided: padded weight arrays, a row vector, a transposing copy of the centres -- what the TRAIN skinning
    field repacks every step -- equal to Tensor.copy_ bit for bit."""
    from vidu4d_amd.lab4d.lbs_fused import copy_strided
    dev = gpu_device
    g = torch.Generator().manual_seed(9)
    N = 12_345
    srcs = [torch.randn(64, 75, generator=g), torch.randn(25, 64, generator=g), torch.randn(1, 25, generator=g),
            torch.randn(N, 3, generator=g).t(), torch.randn(64, 64, generator=g), torch.randn(1, 64, generator=g)] + \
           [torch.randn(3, 5, generator=g) for _ in range(5)]
    srcs = [s.to(dev) if s.is_contiguous() else s.t().contiguous().to(dev).t() for s in srcs]
    bufs = [torch.zeros(64, 96, device=dev), torch.zeros(32, 64, device=dev), torch.zeros(1, 32, device=dev),
            torch.ones(4, N, device=dev), torch.zeros(64, 64, device=dev), torch.zeros(1, 64, device=dev)] + \
           [torch.zeros(4, 8, device=dev) for _ in range(5)]
    want = [b.clone() for b in bufs]
    views = lambda bs: [bs[0][:, :75], bs[1][:25], bs[2][:, :25], bs[3][:3], bs[4], bs[5]] + [b[:3, :5] for b in bs[6:]]  # noqa: E731
    for s, d in zip(srcs, views(want)):
        d.copy_(s)
    copy_strided(list(zip(srcs, views(bufs))))
    for a, b in zip(bufs, want):
        assert torch.equal(a, b)
