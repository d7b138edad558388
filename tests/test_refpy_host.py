"""Host-side rows of the hot path against the IMPORTED reference (fixtures written by
tests/golden/make_refpy_golden.py from /root/reference's own Python, CPU): quaternion algebra, KCamera,
render() post-processing, GaussianModel densify / prune / optimizer surgery, Stage-3 loss reduction.
No expected value in this file comes from vidu4d_amd."""
import os
import types

import numpy as np
import pytest
import torch

from vidu4d_amd.lab4d import quat_transform as qt

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name, dev="cpu"):
    return {k: torch.from_numpy(v).to(dev) if v.dtype.kind in "fiub" else v
            for k, v in np.load(os.path.join(G, name)).items()}


def close(a, b, rtol=2e-5, atol=2e-6, what=""):
    a, b = torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.allclose(a.float(), b.float(), rtol=rtol, atol=atol), \
        f"{what}: max abs err {(a.float() - b.float()).abs().max().item():.3e} (scale {b.float().abs().max().item():.3e})"


# ---------------------------------------------------------------------------------------------
def test_quaternion_algebra_torch_path():
    r = load("refpy_quat.npz")
    close(qt.quaternion_mul(r["a4"], r["b4"]), r["mul44"], what="mul44")
    close(qt.quaternion_mul(r["a3"], r["b4"]), r["mul34"], what="mul34")
    close(qt.quaternion_mul(r["a4"], r["b3"]), r["mul43"], what="mul43")
    close(qt.quaternion_conjugate(r["a4"]), r["conj"], what="conj")
    close(qt.axis_angle_to_quaternion(r["aa"]), r["aa_quat"], what="axis_angle")
    close(qt.quaternion_to_matrix(r["qn"]), r["q_matrix"], what="q->R")
    close(qt.matrix_to_quaternion(r["q_matrix"]), r["matrix_q"], what="R->q")
    close(qt.quaternion_apply(r["qn"], r["b3"]), r["q_apply"], what="apply", atol=1e-5)
    close(qt.quaternion_translation_apply(r["qn"], r["t1"], r["b3"]), r["qt_apply"], what="qt_apply", atol=1e-5)
    qi, ti = qt.quaternion_translation_inverse(r["qn"], r["t1"])
    close(qi, r["qt_inv_q"]), close(ti, r["qt_inv_t"])
    qm, tm = qt.quaternion_translation_mul((r["qn"], r["t1"]), (r["pn"], r["t2"]))
    close(qm, r["qt_mul_q"]), close(tm, r["qt_mul_t"])
    dq1 = qt.quaternion_translation_to_dual_quaternion(r["qn"], r["t1"])
    dq2 = qt.quaternion_translation_to_dual_quaternion(r["pn"], r["t2"])
    close(dq1[0], r["dq1_r"]), close(dq1[1], r["dq1_d"])
    q, t = qt.dual_quaternion_to_quaternion_translation(dq1)
    close(q, r["dq_to_q"]), close(t, r["dq_to_t"])
    m = qt.dual_quaternion_mul(dq1, dq2)
    close(m[0], r["dq_mul_r"]), close(m[1], r["dq_mul_d"])
    inv = qt.dual_quaternion_inverse(dq1)
    close(inv[0], r["dq_inv_r"]), close(inv[1], r["dq_inv_d"])
    close(qt.dual_quaternion_apply(dq1, r["b3"]), r["dq_apply"], atol=1e-5)
    close(qt.quaternion_translation_to_se3(r["qn"], r["t1"]), r["se3"])


# ---------------------------------------------------------------------------------------------
def _cameras(dev="cpu"):
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    r = load("refpy_camera.npz", dev)
    holder = types.SimpleNamespace()
    holder.__dict__["_camera_cache"] = {}
    cams = DeformableSurfels.get_gs_Kcamera(holder, r["Kinv"].to(dev), r["H"].tolist(), r["W"].tolist())
    return r, cams


def check_kcamera(dev):
    r, cams = _cameras(dev)
    for i, c in enumerate(cams):
        close(c.FoVx, r[f"c{i}_FoVx"], what="FoVx", rtol=1e-6, atol=0)
        close(c.FoVy, r[f"c{i}_FoVy"], what="FoVy", rtol=1e-6, atol=0)
        close(c.world_view_transform, r[f"c{i}_world_view_transform"], rtol=0, atol=0)
        close(c.projection_matrix, r[f"c{i}_projection_matrix"], rtol=1e-6, atol=1e-7)
        close(c.full_proj_transform, r[f"c{i}_full_proj_transform"], rtol=1e-6, atol=1e-7)
        close(c.camera_center, r[f"c{i}_camera_center"], rtol=0, atol=0)
        assert (c.image_height, c.image_width) == (int(r["H"][i]), int(r["W"][i]))
        assert c.world_view_transform.device.type == torch.device(dev).type


def test_kcamera_from_intrinsics():
    check_kcamera("cpu")


# ---------------------------------------------------------------------------------------------
class _PresetRasterizer(torch.nn.Module):
    """Stands in for the HIP rasterizer so that render()'s own arithmetic can be checked on the CPU."""
    preset, seen = {}, {}

    def __init__(self, raster_settings):
        super().__init__()
        _PresetRasterizer.seen = raster_settings

    def forward(self, **kw):
        p = _PresetRasterizer.preset
        return p["color"], p["radii"], p["allmap"]


RENDER_KEYS = ("acc", "rend_normal", "rend_dist", "surf_depth", "render_depth_median", "render_depth_expected",
               "surf_normal")


RENDER_CASES = [(1, 0.0), (2, 0.0), (2, 1.0), (2, 0.3)]


def check_render(ci, ratio, monkeypatch, dev="cpu", fused_post=False):
    """render() around a preset rasterizer output: the torch chain (CPU or GPU) or the fused HIP kernels
    (csrc/post.hip, fused_post=True on a GPU)."""
    from vidu4d_amd.gs import gaussian_renderer as gr
    r = load("refpy_render.npz", dev)
    _, cams = _cameras(dev)
    cam = cams[ci]
    monkeypatch.setattr(gr, "GaussianRasterizer", _PresetRasterizer)
    allmap = r[f"c{ci}_allmap"].clone().requires_grad_(True)
    H, W = allmap.shape[1:]
    N = r[f"c{ci}_radii"].shape[0]
    _PresetRasterizer.preset = dict(color=torch.zeros(3, H, W, device=dev), radii=r[f"c{ci}_radii"], allmap=allmap)
    pc = types.SimpleNamespace(get_xyz=torch.zeros(N, 3, device=dev, requires_grad=True),
                               get_opacity=torch.zeros(N, 1, device=dev), get_scaling=torch.ones(N, 2, device=dev),
                               get_rotation=torch.ones(N, 4, device=dev),
                               get_features=torch.zeros(N, 16, 3, device=dev), active_sh_degree=2)
    pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=ratio, debug=False,
                                 fused_post=fused_post)
    out = gr.render(cam, pc, pipe, torch.zeros(3, device=dev))
    tag = f"c{ci}_r{int(ratio * 10)}"
    for k in RENDER_KEYS:
        close(out[k], r[f"{tag}_{k}"], what=k, rtol=1e-4, atol=1e-5)
    assert torch.equal(out["visibility_filter"], r[f"{tag}_visibility_filter"].bool())
    (g,) = torch.autograd.grad(sum((out[k] * r[f"c{ci}_G_{k}"]).sum() for k in RENDER_KEYS), allmap)
    # (the reference's own gradient is NaN on planes 0 / 1 of pixels with alpha == 0: 0 * d(x/0); such pixels have
    # no contributor, so the rasterizer backward never reads it.  The torch chain reproduces the NaNs.)
    ref_g = r[f"{tag}_g_allmap"]
    assert not (torch.isnan(g) & ~torch.isnan(ref_g)).any(), "NaN where the reference's gradient is finite"
    g = torch.where(torch.isnan(ref_g), ref_g, g)
    close(torch.nan_to_num(g), torch.nan_to_num(ref_g), what="g_allmap", rtol=1e-3, atol=1e-3)
    # what the rasterizer is handed (a15: GaussianRasterizationSettings as upstream builds it)
    s = _PresetRasterizer.seen
    # the fp32 tan of the fp32 FoV (1 ulp: atan / tan differ between libm builds and between CPU and GPU)
    assert abs(s.tanfovx - r[f"c{ci}_set_tanfovx"].item()) <= 1.2e-7 * abs(s.tanfovx), "tanfovx must be the fp32 tan"
    assert abs(s.tanfovy - r[f"c{ci}_set_tanfovy"].item()) <= 1.2e-7 * abs(s.tanfovy)
    assert [s.image_height, s.image_width] == r[f"c{ci}_set_hw"].tolist()
    close(s.viewmatrix, r[f"c{ci}_set_viewmatrix"], rtol=0, atol=0)
    close(s.projmatrix, r[f"c{ci}_set_projmatrix"], rtol=1e-6, atol=1e-7)
    close(s.campos, r[f"c{ci}_set_campos"], rtol=0, atol=0)
    assert s.sh_degree == int(r[f"c{ci}_set_sh_degree"])


@pytest.mark.parametrize("ci,ratio", RENDER_CASES)
def test_render_postprocessing_torch_chain(ci, ratio, monkeypatch):
    check_render(ci, ratio, monkeypatch)


def check_depth_to_normal(dev):
    from vidu4d_amd.gs.point_utils import depth_to_normal
    r = load("refpy_render.npz", dev)
    _, cams = _cameras(dev)
    for ci in (1, 2):
        d = r[f"c{ci}_d2n_depth"].clone().requires_grad_(True)
        n = depth_to_normal(cams[ci], d)
        close(n, r[f"c{ci}_d2n_normal"], rtol=1e-4, atol=1e-5)
        (g,) = torch.autograd.grad((n * r[f"c{ci}_d2n_G"]).sum(), d)
        close(g, r[f"c{ci}_d2n_g_depth"], rtol=1e-3, atol=1e-3 * float(r[f"c{ci}_d2n_g_depth"].abs().max()))


def test_depth_to_normal():
    check_depth_to_normal("cpu")


# ---------------------------------------------------------------------------------------------
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "regist_feat")


def _params(gm):
    return {g["name"]: g["params"][0] for g in gm.optimizer.param_groups}


def check_densify(dev, fused=False):
    from vidu4d_amd.gs.gaussian_model import GaussianModel
    r = load("refpy_densify.npz", dev)
    gm = GaussianModel(3, device=dev)
    torch.manual_seed(5)
    pcd = types.SimpleNamespace(points=r["pcd_points"].cpu().numpy(), colors=r["pcd_colors"].cpu().numpy())
    gm.create_from_pcd(pcd, 1.0)
    for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_opacity"):
        close(getattr(gm, k), r["init" + k], what="init" + k, rtol=1e-5, atol=1e-6)
    assert gm._rotation.shape == r["init_rotation"].shape and float(gm._rotation.min()) >= 0  # uniform random (:141)
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=5e-5, position_lr_final=5e-7,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000)
    gm.training_setup(args)
    close(torch.tensor([gm.xyz_scheduler_args(s) for s in (0, 1, 100, 5000, 30000)]), r["lr_sched"], rtol=1e-6)

    # the reference's state right before densification (parameters + Adam moments after two steps)
    P = {n: torch.nn.Parameter(r[f"pre_{n}"].clone()) for n in GROUPS + ("bg_rgb",)}
    gm._xyz, gm._features_dc, gm._features_rest = P["xyz"], P["f_dc"], P["f_rest"]
    gm._opacity, gm._scaling, gm._rotation, gm._regist_feat = P["opacity"], P["scaling"], P["rotation"], P["regist_feat"]
    lr = dict(xyz=5e-5, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3, regist_feat=2.5e-3,
              bg_rgb=2.5e-3)
    if fused:  # the HIP optimizer + device-side densification (csrc/optim.hip)
        from vidu4d_amd.gs.surfel_optim import SurfelAdam as Adam
    else:
        Adam = torch.optim.Adam
    gm.optimizer = Adam([{"params": [P[n]], "lr": lr[n], "name": n} for n in GROUPS + ("bg_rgb",)], lr=5e-4, eps=1e-15)
    for n, p in P.items():
        gm.optimizer.state[p] = {"step": torch.tensor(2.0), "exp_avg": r[f"pre_m_{n}"].clone(),
                                 "exp_avg_sq": r[f"pre_v_{n}"].clone()}
    N = P["xyz"].shape[0]
    gm.xyz_gradient_accum, gm.denom = torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev)
    for f in range(2):
        gm.add_densification_stats(types.SimpleNamespace(grad=r[f"stats{f}_grad"]), r[f"stats{f}_filter"].bool())
    close(gm.xyz_gradient_accum, r["stats_accum"], rtol=1e-6, atol=1e-9)
    close(gm.denom, r["stats_denom"], rtol=0, atol=0)
    gm.max_radii2D = r["pre_max_radii2D"].clone()

    if fused:
        from vidu4d_amd.gs.surfel_optim import densify_and_prune_fused
        densify_and_prune_fused(gm, 2e-4, 0.005, 1.0, 20, samples=r["split_samples"])
    else:
        gm.densify_and_prune(2e-4, 0.005, 1.0, 20, samples=r["split_samples"])
    after = _params(gm)
    for n in GROUPS + ("bg_rgb",):
        close(after[n], r[f"post_{n}"], what="post_" + n, rtol=1e-5, atol=1e-6)
        st = gm.optimizer.state[after[n]]
        close(st["exp_avg"], r[f"post_m_{n}"], what="m_" + n, rtol=1e-6, atol=1e-12)
        close(st["exp_avg_sq"], r[f"post_v_{n}"], what="v_" + n, rtol=1e-6, atol=1e-15)
    assert float(gm.optimizer.state[after["xyz"]]["step"]) == float(r["post_step_xyz"])
    close(gm.xyz_gradient_accum, r["post_accum"]), close(gm.denom, r["post_denom"])
    close(gm.max_radii2D, r["post_max_radii2D"])
    assert gm._xyz is after["xyz"] and gm._regist_feat is after["regist_feat"]

    gm.reset_opacity()
    after = _params(gm)
    close(after["opacity"], r["reset_opacity"], rtol=1e-6, atol=1e-6)
    # upstream files the zeroed moments under the OLD parameter: the new opacity parameter has no Adam state
    assert len(gm.optimizer.state.get(after["opacity"], {})) == int(r["reset_state_len"]) == 0
    for n, p in after.items():
        p.grad = r[f"step2_grad_{n}"].clone()
    gm.optimizer.step()
    close(_params(gm)["opacity"], r["step2_opacity"], what="opacity after the restarted Adam step", rtol=1e-5, atol=1e-6)
    close(_params(gm)["scaling"], r["step2_scaling"], rtol=1e-5, atol=1e-6)

    gm.prune_points(r["prune_mask"].bool())
    close(_params(gm)["xyz"], r["pruned_xyz"], rtol=1e-5, atol=1e-6)
    close(gm.optimizer.state[_params(gm)["xyz"]]["exp_avg"], r["pruned_m_xyz"], rtol=1e-5, atol=1e-12)


def test_gaussian_model_densify_prune_surgery():
    check_densify("cpu")


# ---------------------------------------------------------------------------------------------
LOSS_CASES = ["plain", "undetected", "early", "empty_mask"]


def check_losses(case, dev):
    from vidu4d_amd.lab4d.deformable_surfels import _Args
    from vidu4d_amd.lab4d.stage3 import compute_losses
    r = load("refpy_losses.npz", dev)
    cfg = _Args(dict(lambda_normal=0.05, lambda_dist=100.0, lambda_dssim=0.0, rgb_wt=0.1, mask_wt=0.1))
    leaves = {k: r[f"{case}_in_{k}"].clone().requires_grad_(True) for k in ("rendered", "mask", "rend_dist",
                                                                            "rend_normal", "surf_normal")}
    batch = {"rgb": r[f"{case}_batch_rgb"], "mask": r[f"{case}_batch_mask"], "vis2d": r[f"{case}_batch_vis2d"],
             "is_detected": r[f"{case}_batch_is_detected"]}
    losses = compute_losses(leaves, batch, int(r[f"{case}_step"]), cfg)
    for k in ("rgb", "mask", "normal_loss", "dist_loss"):
        close(losses[k], r[f"{case}_loss_{k}"], what=f"{case}:{k}", rtol=1e-5, atol=1e-8)
    grads = torch.autograd.grad(sum(losses.values()), list(leaves.values()), allow_unused=True)
    for k, g in zip(leaves, grads):
        g = torch.zeros_like(leaves[k]) if g is None else g
        close(g, r[f"{case}_g_{k}"], what=f"{case}:g_{k}", rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("case", LOSS_CASES)
def test_stage3_losses(case):
    check_losses(case, "cpu")


def test_ply_bytes_match_the_reference_writer(tmp_path):
    """gs/scene/gaussian_model.py:189-220 through the imported reference (make_refpy_golden.py::gen_ply): the product's
    save_ply writes the SAME FILE -- attribute list and order (x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*),
    channel-major flattening of the SH tensors, zero normals, float32 little-endian records -- and load_ply reads the
    reference's file back into the reference's tensors."""
    import numpy as np
    from vidu4d_amd.gs.gaussian_model import GaussianModel
    r = np.load(os.path.join(G, "refpy_ply.npz"))
    m = GaussianModel(3, device="cpu")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        setattr(m, k, torch.nn.Parameter(torch.tensor(r["in" + k])))
    assert m.construct_list_of_attributes() == [str(a) for a in r["attributes"]]
    path = str(tmp_path / "out" / "point_cloud.ply")
    m.save_ply(path)
    want = r["file_bytes"].tobytes()
    got = open(path, "rb").read()
    assert got[:got.index(b"end_header\n")] == want[:want.index(b"end_header\n")], "PLY header differs from the reference's"
    assert got == want
    ref_path = str(tmp_path / "ref.ply")
    open(ref_path, "wb").write(want)
    m2 = GaussianModel(3, device="cpu")
    m2.load_ply(ref_path)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert np.array_equal(getattr(m2, k).detach().numpy(), r["in" + k]), k
