"""SURVEY 8 a17, the per-frame render loop, against the IMPORTED reference: `tests/golden/refpy_loop.npz` holds what the
reference's own `DeformableGaussian.query_field` / `render_view` (deformable_gaussian.py:1048-1275, :178-202) produce
around `tests/golden/fake_raster.py` -- the (M,H,W,C) maps, what every frame's rasterizer call is handed (warped centres,
activated rotations / scales / opacities, SH rows, settings), the per-frame densification handles and the gradients
w.r.t. the canonical surfels and the learnable background.  `DeformableSurfels.render_frames` runs around the same
function.  No expected value in this file comes from vidu4d_amd."""
import importlib.util
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("rendered", "mask", "rend_dist", "rend_normal", "surf_normal", "surf_depth", "render_depth_median",
        "render_depth_expected")
LEAVES = ("xyz", "rotation", "scaling", "opacity", "features_dc", "features_rest")


def _fake_raster():
    spec = importlib.util.spec_from_file_location("_vidu4d_fake_raster", os.path.join(G, "fake_raster.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fake_raster


def close(a, b, rtol=1e-4, atol=2e-6, what=""):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).detach().cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), \
        f"{what}: max abs err {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e})"


def build_field(r, learnable_bg, dev="cpu", **opts):
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.nets import make_frame_info
    nets = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)["v2"]
    fi = make_frame_info(nets["offsets"])
    m = DeformableSurfels(dict(fg_motion="gs-bob", gs_learnable_bg=learnable_bg, sh_degree=3, **opts),
                          int(nets["offsets"][-1]), device=dev, data_info={"frame_info": fi, "rtmat": nets["rtmat"]})
    m.warp.load_state_dict(nets["warp"], strict=True)
    m.camera_mlp.load_state_dict(nets["camera_mlp"], strict=True)
    m.warp.eval()
    for n in LEAVES:
        setattr(m, "_" + n, nn.Parameter(r["in_" + n].clone().to(dev)))
    m._regist_feat = nn.Parameter(torch.zeros(r["in_xyz"].shape[0], 16, device=dev))
    m.active_sh_degree = 2
    if learnable_bg:
        m.learnable_bkgd = nn.Parameter(r["in_learnable_bkgd"].clone().to(dev))
    return m.to(dev)


@pytest.mark.parametrize("tag", ["bg", "nobg"])
def test_render_loop_equals_reference_query_field(tag, monkeypatch):
    from vidu4d_amd.gs import gaussian_renderer as gr
    r = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "refpy_loop.npz")).items()}
    fake = _fake_raster()
    calls = []

    class Recorder(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            assert colors_precomp is None and cov3D_precomp is None
            calls.append(dict(settings=self.raster_settings, means3D=means3D, means2D=means2D, opacities=opacities,
                              shs=shs, scales=scales, rotations=rotations))
            return fake(self.raster_settings, means3D, opacities, shs, scales, rotations)

    monkeypatch.setattr(gr, "GaussianRasterizer", Recorder)
    m = build_field(r, tag == "bg")
    M = r["frame_id"].shape[0]
    samples = {"field2cam": (r["cam_q"], r["cam_t"]), "t_articulation": (r["t_art_r"], r["t_art_d"]),
               "rest_articulation": (r["rest_art_r"], r["rest_art_d"])}
    out = m.render_frames(r["frame_id"], r["Kinv"], r["H"].tolist(), r["W"].tolist(), inst_id=r["inst_id"],
                          samples_dict=samples)
    # what each frame's rasterizer call was handed
    assert len(calls) == M
    for i, c in enumerate(calls):
        s = c["settings"]
        close(c["means3D"], r[f"{tag}_f{i}_means3D"], what=f"means3D[{i}]", rtol=2e-5, atol=2e-6)
        close(c["rotations"], r[f"{tag}_f{i}_rotations"], what=f"rotations[{i}]", rtol=2e-5, atol=2e-6)
        close(c["scales"], r[f"{tag}_f{i}_scales"], what="scales", rtol=1e-6, atol=1e-7)   # (exp / sigmoid: 1 ulp between hosts)
        close(c["opacities"], r[f"{tag}_f{i}_opacities"], what="opacities", rtol=1e-6, atol=1e-7)
        close(c["shs"], r[f"{tag}_f{i}_shs"], what="shs", rtol=0, atol=0)
        assert c["means2D"].shape == r[f"{tag}_f{i}_means2D"].shape and not c["means2D"].any()
        assert c["means2D"].requires_grad                      # .grad of it carries the densification statistic
        tan = r[f"{tag}_f{i}_tanfov"]
        assert abs(float(s.tanfovx) - tan[0].item()) <= 1.2e-7 * tan[0].item()
        assert abs(float(s.tanfovy) - tan[1].item()) <= 1.2e-7 * tan[1].item()
        assert [int(s.image_height), int(s.image_width)] == r[f"{tag}_f{i}_hw"].tolist()
        close(s.bg, r[f"{tag}_f{i}_bg"], what="bg", rtol=0, atol=0)
        close(s.viewmatrix, r[f"{tag}_f{i}_viewmatrix"], rtol=0, atol=0)
        close(s.projmatrix, r[f"{tag}_f{i}_projmatrix"], rtol=1e-6, atol=1e-7)
        assert int(s.sh_degree) == int(r[f"{tag}_f{i}_sh_degree"])
        # the densification handles kept per frame (:1231-1233)
        assert torch.equal(torch.as_tensor(m._radii_batch[i]), r[f"{tag}_f{i}_radii"])
        assert torch.equal(torch.as_tensor(m._visibility_filter_batch[i]), r[f"{tag}_f{i}_visibility_filter"].bool())
        assert list(m._viewspace_points_batch[i].shape) == r[f"{tag}_f{i}_viewspace_shape"].tolist()
    assert not hasattr(m, "_override_xyz") and not hasattr(m, "_override_rotation")
    # the (M,H,W,C) maps and their gradients
    for k in KEYS:
        close(out[k], r[f"{tag}_{k}"], what=k)
    leaves = [getattr(m, "_" + n) for n in LEAVES] + ([m.learnable_bkgd] if tag == "bg" else [])
    names = list(LEAVES) + (["learnable_bkgd"] if tag == "bg" else [])
    grads = torch.autograd.grad(sum((out[k] * r[f"{tag}_G_{k}"]).sum() for k in KEYS), leaves)
    for n, g in zip(names, grads):
        ref = r[f"{tag}_g_{n}"]
        close(g, ref, what="grad " + n, rtol=2e-3, atol=2e-5 * ref.abs().max().item())


@pytest.mark.gpu
def test_trainer_path_hands_the_rasterizer_what_the_reference_loop_does(gpu_device, monkeypatch):
    """The path Stage3Trainer runs (frozen networks: fused HIP warp, `render_frames(outputs=("raw",))`, ONE stacked
    rasterizer call for the frames of the step) hands `rasterize_frames` exactly what the reference's loop hands its M
    rasterizer calls, and the gradients that come back through the HIP warp kernels equal the reference's."""
    import vidu4d_amd.diff_surfel_rasterization as dsr
    dev = gpu_device
    r = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "refpy_loop.npz")).items()}
    fake = _fake_raster()
    seen = {}

    def rasterize_frames(means3D, means2D, shs, opacities, scales, rotations, settings, sh_rest=None, raw_params=False,
                         aux_planes=0):
        # the trainer hands over the canonical parameters (the kernels activate them: test_gpu_parity.py::
        # test_canonical_parameters_equal_the_activated_path); here they are activated as upstream does it
        if sh_rest is not None:
            shs = torch.cat((shs, sh_rest), dim=1)
        if raw_params:
            opacities, scales = torch.sigmoid(opacities), torch.exp(scales)
        seen.update(means3D=means3D, means2D=means2D, shs=shs, opacities=opacities, scales=scales, rotations=rotations,
                    settings=settings, canonical=(sh_rest is not None, bool(raw_params)))
        outs = [fake(s, means3D[i], opacities, shs, scales, rotations[i]) for i, s in enumerate(settings)]
        return (torch.stack([o[0] for o in outs], 1), torch.stack([o[1] for o in outs], 0),
                torch.stack([o[2] for o in outs], 1))

    monkeypatch.setattr(dsr, "rasterize_frames", rasterize_frames)
    m = build_field(r, True, dev=dev)
    for mod in (m.warp, m.camera_mlp):
        for p in mod.parameters():
            p.requires_grad_(False)
    fid, iid = r["frame_id"].to(dev), r["inst_id"].to(dev)
    assert m.fused_warp_ok(iid)
    # frames 0 and 1 share a camera and an image size; the fixture's third frame has its own intrinsics
    out = m.render_frames(fid, r["Kinv"], r["H"].tolist(), r["W"].tolist(), inst_id=iid, outputs=("raw",))
    color, allmap = out["raw_stacked"]
    M = fid.shape[0]
    assert color.shape[:2] == (3, M) and allmap.shape[:2] == (8, M)
    for i in range(M):
        s = seen["settings"][i]
        close(seen["means3D"][i], r[f"bg_f{i}_means3D"], what=f"means3D[{i}]", rtol=2e-5, atol=1e-5)
        close(seen["rotations"][i], r[f"bg_f{i}_rotations"], what=f"rotations[{i}]", rtol=2e-5, atol=1e-5)
        tan = r[f"bg_f{i}_tanfov"]
        assert abs(float(s.tanfovx) - tan[0].item()) <= 1.2e-7 * tan[0].item()
        assert abs(float(s.tanfovy) - tan[1].item()) <= 1.2e-7 * tan[1].item()
        assert [int(s.image_height), int(s.image_width)] == r[f"bg_f{i}_hw"].tolist()
        close(s.bg, r[f"bg_f{i}_bg"], rtol=0, atol=0)
        close(s.viewmatrix, r[f"bg_f{i}_viewmatrix"], rtol=0, atol=0)
        close(s.projmatrix, r[f"bg_f{i}_projmatrix"], rtol=1e-6, atol=1e-7)
        assert int(s.sh_degree) == int(r[f"bg_f{i}_sh_degree"])
        assert torch.equal(torch.as_tensor(m._radii_batch[i]).cpu(), r[f"bg_f{i}_radii"])
        assert torch.equal(torch.as_tensor(m._visibility_filter_batch[i]).cpu(), r[f"bg_f{i}_visibility_filter"].bool())
    close(seen["scales"], r["bg_f0_scales"], rtol=1e-6, atol=1e-7)
    close(seen["opacities"], r["bg_f0_opacities"], rtol=1e-6, atol=1e-7)
    close(seen["shs"], r["bg_f0_shs"], rtol=0, atol=0)
    assert seen["means2D"].shape == seen["means3D"].shape and seen["means2D"].requires_grad
    assert seen["canonical"] == (True, True)
    # the raw planes are the reference's maps before its learnable-background composite / permute / cat: rebuild the two
    # maps that need nothing else (mask = alpha plane, rend_dist = plane 6) and the composite
    close(allmap[1].unsqueeze(-1), r["bg_mask"], what="mask", rtol=1e-4, atol=1e-5)
    close(allmap[6].unsqueeze(-1), r["bg_rend_dist"], what="rend_dist", rtol=1e-4, atol=1e-5)
    comp = color + (1 - allmap[1:2]) * m.learnable_bkgd[:, None, None, None]
    close(comp.permute(1, 2, 3, 0), r["bg_rendered"], what="rendered", rtol=1e-4, atol=1e-5)
    # gradients through the composite and the HIP warp against the reference's (only the three maps above)
    G3 = {k: r[f"bg_G_{k}"].to(dev) for k in ("rendered", "mask", "rend_dist")}
    leaves = [getattr(m, "_" + n) for n in LEAVES] + [m.learnable_bkgd]
    loss = (comp.permute(1, 2, 3, 0) * G3["rendered"]).sum() + (allmap[1].unsqueeze(-1) * G3["mask"]).sum() \
        + (allmap[6].unsqueeze(-1) * G3["rend_dist"]).sum()
    grads = torch.autograd.grad(loss, leaves)
    for n, g in zip(list(LEAVES) + ["learnable_bkgd"], grads):
        ref = r[f"bg3_g_{n}"]
        close(g, ref, what="grad " + n, rtol=5e-3, atol=5e-5 * ref.abs().max().item())
