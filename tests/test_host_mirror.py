"""CPU tests of the host-side mirror of the reference interface: bob LBS warp, dual-quaternion
algebra, KCamera, GaussianModel densify / prune / optimizer surgery, PLY round trip."""
import math

import numpy as np
import pytest
import torch

from vidu4d_amd.gs.cameras import KCamera
from vidu4d_amd.gs.gaussian_model import GaussianModel, build_rotation
from vidu4d_amd.lab4d import quat_transform as qt
from vidu4d_amd.lab4d.bob_warp import SkinningWarp, apply_qt_to_gaussian, dual_quaternion_skinning_qt
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels, PointCloud, _Args
from vidu4d_amd.lab4d.stage3 import Stage3Trainer


def _rand_dq(M, B, g):
    q = torch.nn.functional.normalize(torch.randn(M, B, 4, generator=g), dim=-1)
    t = torch.randn(M, B, 3, generator=g) * 0.3
    return qt.quaternion_translation_to_dual_quaternion(q, t)


def test_dual_quaternion_algebra():
    g = torch.Generator().manual_seed(0)
    a, b = _rand_dq(3, 5, g), _rand_dq(3, 5, g)
    p = torch.randn(3, 5, 3, generator=g)
    # composition == sequential application; inverse undoes
    ab = qt.dual_quaternion_mul(a, b)
    assert torch.allclose(qt.dual_quaternion_apply(ab, p), qt.dual_quaternion_apply(a, qt.dual_quaternion_apply(b, p)),
                          atol=1e-5)
    assert torch.allclose(qt.dual_quaternion_apply(qt.dual_quaternion_inverse(a), qt.dual_quaternion_apply(a, p)), p,
                          atol=1e-5)
    # quaternion_apply == rotation matrix
    q = torch.nn.functional.normalize(torch.randn(7, 4, generator=g), dim=-1)
    v = torch.randn(7, 3, generator=g)
    assert torch.allclose(qt.quaternion_apply(q, v), torch.einsum("nij,nj->ni", build_rotation(q), v), atol=1e-5)


def test_bob_warp_rigid_when_bones_move_together():
    """If every bone undergoes the same rigid motion, every point undergoes exactly that motion."""
    from vidu4d_amd.lab4d.nets import make_frame_info
    torch.manual_seed(3)
    warp = SkinningWarp(make_frame_info([0, 8]), num_se3=25)
    g = torch.Generator().manual_seed(2)
    M, N = 2, 500
    xyz = torch.randn(M, N, 1, 3, generator=g) * 0.1
    _, rest = warp.articulation.get_vals_and_mean(torch.arange(M))
    rest = (rest[0].detach(), rest[1].detach())
    q = torch.nn.functional.normalize(torch.randn(M, 1, 4, generator=g), dim=-1).expand(-1, 25, -1).contiguous()
    t = (torch.randn(M, 1, 3, generator=g) * 0.2).expand(-1, 25, -1).contiguous()
    rigid = qt.quaternion_translation_to_dual_quaternion(q, t)
    t_art = qt.dual_quaternion_mul(rigid, rest)
    (qq, tt), aux = warp(xyz, torch.arange(M), samples_dict={"rest_articulation": rest, "t_articulation": t_art},
                         return_aux=True, return_qt=True)
    moved, _ = apply_qt_to_gaussian(xyz, None, qq, tt, M)
    want = qt.quaternion_translation_apply(q[:, :1], t[:, :1], xyz.view(M, N, 3)).view(M, N, 1, 3)
    assert torch.allclose(moved, want, atol=1e-5)
    assert aux["skin_entropy"].shape == (M, N, 1, 1) and aux["delta_skin"].shape == (M, N, 1, 1)
    # and gradients reach the canonical points through the skinning weights
    xyz.requires_grad_(True)
    (q2, t2) = warp(xyz, torch.arange(M), return_qt=True)
    (q2.sum() + t2.sum()).backward()
    assert xyz.grad is not None and torch.isfinite(xyz.grad).all()


def test_kcamera_matches_reference_conventions():
    H, W = 96, 128
    K = torch.tensor([[W / 1.0, 0, W / 2.0], [0, W / 1.0, H / 2.0], [0, 0, 1.0]])
    Kinv = torch.inverse(K)
    left, right = Kinv[0, 2], Kinv[0, 2] + Kinv[0, 0] * W
    bottom, top = Kinv[1, 2], Kinv[1, 2] + Kinv[1, 1] * H
    cam = KCamera(H=H, W=W, left=left, right=right, top=top, bottom=bottom, data_device="cpu")
    assert torch.equal(cam.world_view_transform, torch.eye(4))
    assert torch.equal(cam.camera_center, torch.zeros(3))
    assert math.isclose(float(torch.tan(cam.FoVx * 0.5)), 0.5, rel_tol=1e-6)
    assert math.isclose(float(torch.tan(cam.FoVy * 0.5)), 0.5 * H / W, rel_tol=1e-6)
    assert cam.full_proj_transform.shape == (4, 4) and cam.original_image.shape == (1, H, W)
    # a point on the optical axis projects to the NDC centre
    p = torch.tensor([0.0, 0.0, 2.0, 1.0]) @ cam.full_proj_transform
    assert abs(float(p[0] / p[3])) < 1e-6 and abs(float(p[1] / p[3])) < 1e-6


def _model(n=400, seed=0):
    rng = np.random.default_rng(seed)
    opts = dict(fg_motion="gs-bob", sh_degree=3)
    m = DeformableSurfels(opts, num_frames=8, device="cpu")
    m.init_from_points(rng.normal(size=(n, 3)).astype(np.float32) * 0.1, rng.uniform(size=(n, 3)).astype(np.float32))
    return m


def test_gaussian_model_init_and_activations():
    m = _model()
    n = m._xyz.shape[0]
    assert m._features_dc.shape == (n, 1, 3) and m._features_rest.shape == (n, 15, 3)
    assert m._scaling.shape == (n, 2) and m._rotation.shape == (n, 4) and m._opacity.shape == (n, 1)
    assert torch.allclose(m.get_opacity, torch.full((n, 1), 0.1), atol=1e-6)
    assert torch.allclose(m.get_rotation.norm(dim=1), torch.ones(n), atol=1e-6)
    assert m.get_features.shape == (n, 16, 3) and m.active_sh_degree == 0
    for _ in range(5):
        m.oneupSHdegree()
    assert m.active_sh_degree == 3
    # scales = sqrt(mean squared distance to the 3 nearest neighbours)
    from scipy.spatial import cKDTree
    pts = m._xyz.detach().numpy()
    d, _ = cKDTree(pts).query(pts, k=4)
    assert np.allclose(m.get_scaling[:, 0].detach().numpy(), np.sqrt((d[:, 1:] ** 2).mean(1)), rtol=1e-4)


def test_densify_prune_keeps_optimizer_state_consistent():
    m = _model(300)
    tr = Stage3Trainer(m, dict(fg_motion="gs-bob"))
    # one fake step so that Adam has state
    for p in tr.surfel_params():
        p.grad = torch.randn_like(p) * 1e-3
    tr.gs_optimizer.step()
    n0 = m._xyz.shape[0]
    m.xyz_gradient_accum = torch.rand(n0, 1) * 1e-3
    m.denom = torch.ones(n0, 1)
    m._opacity.data[:10] = -10.0  # ~0 opacity: pruned
    gen = torch.Generator().manual_seed(5)
    m.densify_and_prune(2e-4, 0.005, extent=1.0, max_screen_size=None, generator=gen)
    n1 = m._xyz.shape[0]
    assert n1 != n0
    for group in tr.gs_optimizer.param_groups:
        if group["name"] == "bg_rgb":
            continue
        p = group["params"][0]
        assert p.shape[0] == n1 and p.requires_grad
        st = tr.gs_optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert m.xyz_gradient_accum.shape == (n1, 1) and m.denom.shape == (n1, 1) and m.max_radii2D.shape == (n1,)
    assert m._regist_feat.shape[0] == n1
    assert float(m.get_opacity.min()) >= 0.005 - 1e-6
    # the optimizer still steps on the new tensors
    for p in tr.surfel_params():
        p.grad = torch.zeros_like(p)
    tr.gs_optimizer.step()
    m.reset_opacity()
    assert float(m.get_opacity.max()) <= 0.01 + 1e-6


def test_ply_round_trip(tmp_path):
    m = _model(50)
    path = str(tmp_path / "fg-gs.ply")
    m.save_ply(path)
    head = open(path, "rb").read(400).decode("ascii", "ignore")
    assert "property float x" in head and "property float nx" in head and "property float f_dc_0" in head
    m2 = GaussianModel(3, device="cpu")
    m2.load_ply(path)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(m, k).detach(), getattr(m2, k).detach()), k
    names = m.construct_list_of_attributes()
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[-4:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(names) == 6 + 3 + 45 + 1 + 2 + 4


def test_stage3_defaults_match_reference_flags():
    a = _Args({})
    assert (a.position_lr_init, a.feature_lr, a.opacity_lr, a.scaling_lr, a.rotation_lr) == (5e-5, 2.5e-3, 0.05, 5e-3, 1e-3)
    assert (a.densification_interval, a.densify_from_iter, a.densify_until_iter) == (100, 500, 15000)
    assert (a.densify_grad_threshold, a.opacity_reset_interval, a.percent_dense) == (2e-4, 3000, 0.01)


def test_train_entry_flag_parsing(tmp_path):
    """The reference's Stage-3 command line (README.md:44) parses; foreign flags are tolerated."""
    from vidu4d_amd.lab4d.train import load_obj_points, parse_flags
    argv = ("--seqname cat-pikachu-0 --logname gs --fg_motion gs-bob --num_rounds 61 --load_path a/ckpt_0020.pth "
            "--gs_init_mesh a/021-fg-geo.obj --imgs_per_gpu 1 --pixels_per_image -1 --eval_res 256 --rgb_timefree "
            "--rgb_dirfree --rgb_loss_only --gs_optim_warp=False --data_prefix full --force_center_cam "
            "--nogs_learnable_bg --feature_lr=0.001").split()
    opts, ignored = parse_flags(argv)
    assert opts["fg_motion"] == "gs-bob" and opts["num_rounds"] == 61 and opts["pixels_per_image"] == -1
    assert opts["rgb_loss_only"] is True and opts["gs_optim_warp"] is False and opts["force_center_cam"] is True
    assert opts["gs_learnable_bg"] is False and opts["feature_lr"] == 0.001
    assert ignored == ["--rgb_timefree", "--rgb_dirfree"]
    ff = tmp_path / "opts.log"
    ff.write_text("--densification_interval=50\n--sh_degree=2\n")
    opts, _ = parse_flags([f"--flagfile={ff}", "--seqname", "x"])
    assert opts["densification_interval"] == 50 and opts["sh_degree"] == 2 and opts["seqname"] == "x"
    obj = tmp_path / "m.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1/1 2/1 4/1\n")
    pts = load_obj_points(str(obj), 500, np.random.default_rng(0))
    assert pts.shape == (500, 3) and (pts >= -1e-6).all() and (pts.sum(1) <= 1 + 1e-5).all()


def test_checkpoint_layout_roundtrip(tmp_path):
    """ckpt_%04d.pth in the reference layout (trainer.py:335-422): key prefix, DDP prefix stripping,
    surfel tensors resized to the checkpoint's point count, optimizer rebuilt, PLY next to it."""
    import numpy as np
    import torch
    from vidu4d_amd.lab4d import checkpoint as ck
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer

    def make(n, seed):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        m = DeformableSurfels(dict(fg_motion="gs-bob"), num_frames=4, device="cpu")
        m.init_from_points(rng.normal(size=(n, 3)).astype(np.float32) * 0.2, rng.uniform(size=(n, 3)).astype(np.float32))
        return m

    a = make(300, 0)
    ta = Stage3Trainer(a)
    ta.current_steps = 1234
    path = ck.save_checkpoint(ta, str(tmp_path), round_count=3)
    assert path.endswith("ckpt_0003.pth") and (tmp_path / "ckpt_latest.pth").exists() and (tmp_path / "003-fg-gs.ply").exists()
    raw = torch.load(path, weights_only=False)
    assert set(raw) >= {"current_steps", "current_round", "model", "optimizer"} and raw["current_round"] == 3
    for k in ck.SURFEL_KEYS:
        assert ck.FG_PREFIX + k in raw["model"]
    # a DDP-style checkpoint into a model with a different point count
    raw["model"] = {"module." + k: v for k, v in raw["model"].items()}
    torch.save(raw, tmp_path / "ddp.pth")
    b = make(120, 1)
    tb = Stage3Trainer(b)
    info = ck.load_checkpoint(str(tmp_path / "ddp.pth"), b, tb, reset_steps=False)
    assert b._xyz.shape[0] == 300 and b.max_radii2D.shape[0] == 300 and tb.current_steps == 1234
    ck.load_checkpoint(str(tmp_path / "ddp.pth"), b, tb)  # the reference's default: --reset_steps
    assert tb.current_steps == 0
    for k in ck.SURFEL_KEYS:
        assert torch.equal(getattr(a, k).detach(), getattr(b, k).detach())
    assert torch.equal(a.warp.skinning_model.log_gauss.detach(), b.warp.skinning_model.log_gauss.detach())
    assert not info["unexpected_keys"]
    assert tb.gs_optimizer.param_groups[0]["params"][0] is b._xyz


def test_vidloader_reads_reference_layout(tmp_path):
    """database/processed/... .npy layout (vidloader.py:81-166) -> a Stage-3 frame batch; Kinv as
    model.py:417-427 builds it from the raw intrinsics and the crop transform."""
    import numpy as np
    import torch
    from vidu4d_amd.lab4d.vidloader import K2inv, K2mat, SequenceData
    rng = np.random.default_rng(0)
    F, H, W = 5, 12, 16
    seq, prefix = "cat-0000", "full-256"
    for sub in ("JPEGImages", "Annotations"):
        (tmp_path / sub / "Full-Resolution" / seq).mkdir(parents=True)
    rgb = rng.uniform(size=(F, H, W, 3)).astype(np.float16)
    ann = (rng.uniform(size=(F, H, W, 2)) > 0.4)
    crop2raw = np.stack([np.array([2.0, 2.0, 10.0 + i, 20.0]) for i in range(F)]).astype(np.float32)
    np.save(tmp_path / "JPEGImages" / "Full-Resolution" / seq / f"{prefix}.npy", rgb)
    np.save(tmp_path / "Annotations" / "Full-Resolution" / seq / f"{prefix}.npy", ann)
    np.save(tmp_path / "Annotations" / "Full-Resolution" / seq / f"{prefix}-crop2raw.npy", crop2raw)
    np.save(tmp_path / "Annotations" / "Full-Resolution" / seq / f"{prefix}-is_detected.npy", np.array([1, 1, 0, 1, 1]))
    ds = SequenceData(str(tmp_path), seq, prefix)
    assert len(ds) == F and ds.img_size == (H, W)
    K = np.array([500.0, 510.0, 320.0, 240.0], dtype=np.float32)
    b = ds.frame_batch([3, 1], K, frame_offset=100)
    assert b["frameid"].tolist() == [103, 101] and b["H"] == [H, H] and b["W"] == [W, W]
    assert b["rgb"].shape == (2, H, W, 3) and b["rgb"].dtype == torch.float32
    assert np.allclose(b["rgb"].numpy(), rgb[[3, 1]].astype(np.float32))
    assert np.array_equal(b["mask"].numpy()[..., 0] > 0, ann[[3, 1], ..., 0])
    assert np.array_equal(b["vis2d"].numpy()[..., 0] > 0, ann[[3, 1], ..., 1]) and b["is_detected"].tolist() == [True, True]
    # a crop pixel (u, v) maps to the raw pixel (2u + cx_c, 2v + cy_c), then to the ray through it
    u, v = 5.0, 7.0
    ray = b["Kinv"][0] @ torch.tensor([u, v, 1.0])
    assert np.allclose(ray.numpy(), [((2 * u + 13.0) - 320.0) / 500.0, ((2 * v + 20.0) - 240.0) / 510.0, 1.0], atol=1e-6)
    assert torch.allclose(K2inv(torch.tensor(K)) @ K2mat(torch.tensor(K)), torch.eye(3), atol=1e-6)


def test_train_cli_flag_parsing_and_mesh_sampling(tmp_path):
    """lab4d/train.py: absl-style flags of the reference's Stage-3 command line (README.md:44), unknown
    flags collected as ignored, --flagfile expansion; area-weighted sampling of an OBJ proxy mesh."""
    import numpy as np
    from vidu4d_amd.lab4d.train import load_obj_points, parse_flags
    ff = tmp_path / "opts.log"
    ff.write_text("--num_rounds=3\n# comment\n--iters_per_round 50\n")
    opts, ignored = parse_flags(["--seqname", "cat-pikachu-0", "--logname=gs", "--fg_motion", "gs-bob", "--imgs_per_gpu", "1",
                                 "--pixels_per_image", "-1", "--rgb_timefree", "--rgb_dirfree", "--rgb_loss_only",
                                 "--gs_optim_warp=False", "--data_prefix", "full", "--force_center_cam", "--nogs_learnable_bg",
                                 "--eval_res", "256", f"--flagfile={ff}", "--feature_lr", "0.005"])
    assert opts["seqname"] == "cat-pikachu-0" and opts["logname"] == "gs" and opts["fg_motion"] == "gs-bob"
    assert opts["rgb_loss_only"] is True and opts["gs_optim_warp"] is False and opts["force_center_cam"] is True
    assert opts["gs_learnable_bg"] is False and opts["pixels_per_image"] == -1 and opts["eval_res"] == 256
    assert opts["num_rounds"] == 3 and opts["iters_per_round"] == 50 and opts["feature_lr"] == 0.005
    assert "--rgb_timefree" in ignored and "--rgb_dirfree" in ignored
    obj = tmp_path / "proxy.obj"  # a unit square of two triangles plus a far, tiny triangle
    obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 5 5 5\nv 5.001 5 5\nv 5 5.001 5\nf 1 2 3 4\nf 5/1 6/1 7/1\n")
    pts = load_obj_points(str(obj), 4000, np.random.default_rng(0))
    assert pts.shape == (4000, 3) and pts.dtype == np.float32
    on_square = (np.abs(pts[:, 2]) < 1e-6) & (pts[:, 0] >= 0) & (pts[:, 0] <= 1) & (pts[:, 1] >= 0) & (pts[:, 1] <= 1)
    assert on_square.mean() > 0.99                      # area-weighted: the tiny triangle gets ~1e-6 of the samples
    assert abs((pts[on_square, 0] > pts[on_square, 1]).mean() - 0.5) < 0.05   # both halves of the quad


def test_deferred_check_bookkeeping():
    """_C's host-side policy after a deferred forward (no GPU involved): the pair count feeds the capacity hint, an
    overflow or a frame the segment limit cut short (`truncated`) makes the check fail -- the step is replayed -- and the
    next frames of the shape run without the limit."""
    import torch
    from vidu4d_amd import _C
    key = ("test-shape",)
    for d in (_C._unlimited, _C._capacity_hint, _C._len_hint):
        d.pop(key, None)
    slot = torch.zeros(16, dtype=torch.int32)
    slot[0] = 1000
    assert _C.check_slots([(slot, None, 2000, key)]) is True
    assert _C._capacity_hint[key] >= 1000 and key not in _C._unlimited
    assert _C.check_slots([(slot, None, 900, key)]) is False      # more pairs than the buffer held
    slot[6] = 1                                                    # truncated
    assert _C.check_slots([(slot, None, 2000, key)]) is False
    assert _C._unlimited[key] == 4
    for d in (_C._unlimited, _C._capacity_hint, _C._len_hint):
        d.pop(key, None)


def test_long_list_sort_mode_follows_the_longest_list_of_earlier_frames():
    """_C's host-side policy for Vidu4dSurfelForwardArgs::long_list_sort (no GPU involved): word 2 of the header (the longest
    tile list) feeds a slowly decaying maximum per image shape; the MSD split is used from MSD_SORT_FROM entries on and
    for shapes nothing is known about yet."""
    import torch
    from vidu4d_amd import _C
    key = ("test-shape-sort",)
    _C._len_hint.pop(key, None)
    mode = lambda: 0 if _C._len_hint.get(key, 1 << 30) >= _C.MSD_SORT_FROM else 1  # noqa: E731  (as rasterize_gaussians)
    assert mode() == 0                                   # unknown shape: the split (never slower by much)
    slot = torch.zeros(16, dtype=torch.int32)
    slot[0], slot[2] = 1000, 5000
    _C.check_slots([(slot, None, 2000, key)])
    assert _C._len_hint[key] == 5000 and mode() == 1     # a few thousand entries: one workgroup per long list
    slot[2] = 25000
    _C.check_slots([(slot, None, 2000, key)])
    assert mode() == 0                                   # rises at once ...
    slot[2] = 5000
    _C.check_slots([(slot, None, 2000, key)])
    assert _C._len_hint[key] == 22500 and mode() == 0    # ... and decays by a tenth per frame
    for _ in range(10):
        _C.check_slots([(slot, None, 2000, key)])
    assert mode() == 1
    for d in (_C._len_hint, _C._unlimited, _C._capacity_hint):
        d.pop(key, None)


def test_auto_split_wants_deep_blending_and_too_few_long_tiles():
    """_C's host-side policy for Vidu4dSurfelForwardArgs::segment_split under VIDU4D_SURFEL_SPLIT=auto (no GPU involved): the
    forward is segment-parallel only when a pixel of the earlier frames blended deeper than SPLIT_AUTO_LEN AND those frames
    had fewer long tiles (header word 14: tiles longer than 1024 entries, the same count after a whole-tile and after a split
    frame; word 4, which the rule read until round 4, follows the mode) than SPLIT_AUTO_TILES_PER_CU per compute unit -- a frame with enough long tiles
    fills the chip with whole-tile walks, which stop at saturation, and its backward is segment-parallel anyway."""
    import torch
    from vidu4d_amd import _C
    cu = 256
    few, many = int(_C.SPLIT_AUTO_TILES_PER_CU * cu) - 1, int(_C.SPLIT_AUTO_TILES_PER_CU * cu)
    assert not _C.auto_split(_C.SPLIT_AUTO_LEN, few, cu)          # shallow: never
    assert _C.auto_split(_C.SPLIT_AUTO_LEN + 1, few, cu)          # deep, few long tiles: split
    assert not _C.auto_split(10 * _C.SPLIT_AUTO_LEN, many, cu)    # deep, but the chip is full of long tiles already
    assert _C.auto_split(_C.SPLIT_AUTO_LEN + 1, 0, cu)            # nothing known about the tiles yet: the depth decides
    key = ("test-shape-split",)
    _C._long_tiles_hint.pop(key, None)
    slot = torch.zeros(16, dtype=torch.int32)
    slot[0], slot[2], slot[_C.HEADER_LONG_TILES_WORD], slot[4] = 1000, 5000, 720, 9999
    _C.check_slots([(slot, None, 2000, key)])
    assert _C._long_tiles_hint[key] == 720                          # the latest frame's count, no smoothing
    slot[_C.HEADER_LONG_TILES_WORD] = 350
    _C.check_slots([(slot, None, 2000, key)])
    assert _C._long_tiles_hint[key] == 350
    for d in (_C._long_tiles_hint, _C._len_hint, _C._unlimited, _C._capacity_hint):
        d.pop(key, None)


def test_depth_only_sort_with_tie_fix_up_is_the_reference_order():
    """The tile sort's algorithm (csrc/binning.hip), restated in numpy: LSD passes on the four depth bytes of keys that
    arrive in ARBITRARY order (the emission order inside a group is not deterministic), then every run of equal depths
    put in id order (runs longer than the cap fall back to the full (id bytes, depth bytes) sequence) == the reference's
    stable sort by depth of id-ordered entries, i.e. ascending (depth, id) -- for few ties, clone clouds and one long run."""
    rng = np.random.default_rng(5)

    def product_order(keys, run_cap=32):
        k = keys.copy()
        for shift in (32, 40, 48, 56):                       # stable counting passes on the depth bytes
            k = k[np.argsort((k >> np.uint64(shift)) & np.uint64(255), kind="stable")]
        d = (k >> np.uint64(32)).astype(np.uint64)
        starts = np.flatnonzero(np.r_[True, d[1:] != d[:-1]])
        ends = np.r_[starts[1:], len(k)]
        if (ends - starts).max() > run_cap:                  # a long run: the full LSD sequence on the list as it stands
            for shift in (0, 8, 16, 24, 32, 40, 48, 56):
                k = k[np.argsort((k >> np.uint64(shift)) & np.uint64(255), kind="stable")]
            return k
        for a, b in zip(starts, ends):
            if b - a > 1:
                k[a:b] = np.sort(k[a:b])                     # equal depths: the 64-bit keys order by id
        return k

    for n, n_depths in ((5000, 10 ** 9), (5000, 2000), (3000, 40), (900, 1)):
        depth = rng.integers(0x40000000, 0x40000000 + n_depths, size=n, dtype=np.uint64)   # float bits of [2, 4)
        ids = rng.permutation(n).astype(np.uint64)           # distinct ids, arbitrary arrival order
        keys = (depth << np.uint64(32)) | ids
        want = np.sort(keys)                                 # ascending (depth, id)
        assert np.array_equal(product_order(keys), want), (n, n_depths)


def test_relative_segment_blend_algorithm_in_numpy():
    """The algorithm behind assume_unsaturated (csrc/blend.hip), restated for one pixel in fp32 numpy: segments blended
    from T = 1 and scaled by the running product of their predecessors; the first segment whose end would lie within
    0.1 % of the saturation threshold (or below it) is blended again, in order, from that running product, and ends the
    pixel == the sequential front-to-back blend: the same last contributor and, where the pixel saturates, bit for bit
    the same final transmittance whenever the running product is what the sequential blend has at that point."""
    rng = np.random.default_rng(9)
    T_EPS, SEG = np.float32(1e-4), 512

    def sequential(alpha, col, T0=np.float32(1)):
        T, C, last = np.float32(T0), np.zeros(3, np.float32), 0
        for i, (a, c) in enumerate(zip(alpha, col)):
            test_T = np.float32(T * np.float32(1 - a))
            if test_T < T_EPS:
                break
            C = (C + c * np.float32(a * T)).astype(np.float32)
            T, last = test_T, i + 1
        return C, T, last

    def relative(alpha, col):
        T_run, C, last, repaired = np.float32(1), np.zeros(3, np.float32), 0, False
        for s0 in range(0, len(alpha), SEG):
            a_seg, c_seg = alpha[s0:s0 + SEG], col[s0:s0 + SEG]
            C_loc, T_loc, last_loc = sequential(a_seg, c_seg)
            sat = last_loc < len(a_seg)                      # a sample was refused although the walk started from T = 1
            T_end = np.float32(T_run * (np.float32(0) if sat else T_loc))
            if not (T_end >= T_EPS * np.float32(1.001)):     # saturates in here (or nearly): again, from the exact start
                C_abs, T_fin, last_abs = sequential(a_seg, c_seg, T_run)
                C = (C + C_abs).astype(np.float32)
                return C, T_fin, (s0 + last_abs if last_abs else last), True
            C = (C + T_run * C_loc).astype(np.float32)
            if last_loc:
                last = s0 + last_loc
            T_run = T_end
        return C, T_run, last, repaired

    for n, a_max, expect_repair in ((3000, 0.002, False), (1800, 0.004, False), (3000, 0.02, True), (700, 0.9, True)):
        alpha = rng.uniform(0.0, a_max, size=n).astype(np.float32)
        col = rng.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
        C0, T0, last0 = sequential(alpha, col)
        C1, T1, last1, repaired = relative(alpha, col)
        assert repaired == expect_repair, (n, a_max, T0)
        assert last1 == last0 and (repaired or last0 == n)
        assert abs(T1 - T0) <= 2e-6 * T0 and np.abs(C1 - C0).max() <= 2e-6 * np.abs(C0).max()


def test_modelled_scaling_arithmetic():
    """bench.py's MODELLED multi-GPU figures (no node to measure them on): the all-reduce cost model and the two ways a step
    pays for it -- plain arithmetic, pinned here so that the numbers on the bench line mean what BASELINE.md says."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    S = 200_000 * 58 * 4
    assert b.exchange_model_ms(S, 1) == {"ring_one_link": 0.0, "ring_all_links": 0.0, "direct_rs_ag": 0.0}
    e8 = b.exchange_model_ms(S, 8)
    # ring over one link per hop: 2 (n-1)/n S bytes at 153 GB/s + two 20 us phases
    assert abs(e8["ring_one_link"] - (2 * 7 / 8 * S / 153e9 * 1e3 + 0.04)) < 1e-9
    assert e8["direct_rs_ag"] < e8["ring_all_links"] < e8["ring_one_link"]
    # two GPUs have one link between them: the "all links" ring is the one-link ring
    e2 = b.exchange_model_ms(S, 2)
    assert abs(e2["ring_all_links"] - e2["ring_one_link"]) < 1e-12
    serial = b.scaling_model(1.5, S, overlapped=False)["8"]
    over = b.scaling_model(1.5, S, overlapped=True)["8"]
    assert over["ring_all_links"] == 8.0 and 6.0 < serial["ring_all_links"] < 8.0
    assert abs(serial["ring_all_links"] - round(8 * 1.5 / (1.5 + e8["ring_all_links"]), 2)) < 1e-9
    # the fitting step as Stage3Trainer issues its exchange: only what of the SH bands' collective outlasts the warp's backward,
    # plus the small tensors' collective, is serial -- never worse than everything serial, never better than free
    fit = b.fit_scaling_model(1.5, 200_000)["8"]
    for k in ("ring_one_link", "ring_all_links", "direct_rs_ag"):
        assert serial[k] <= fit[k] <= 8.0
    small = b.exchange_model_ms(200_000 * 13 * 4 + 12, 8)["ring_all_links"]
    rest = b.exchange_model_ms(200_000 * 45 * 4, 8)["ring_all_links"]
    assert abs(fit["serial_exchange_ms"]["ring_all_links"] - round(max(0.0, rest - b.WARP_BACKWARD_MS) + small, 3)) < 1e-9


def test_raster_contexts_do_not_share_state():
    """Round 5: the rasterizer's host-side state lives in RasterContext objects the caller owns; the
    module-level API acts on the calling thread's current one.  No GPU involved."""
    import threading
    import torch
    from vidu4d_amd import _C, _lib
    assert _C.current() is _C._default and _C._capacity_hint is _C._default.capacity_hint and _C._pending is _C._default.pending
    a, b = _C.RasterContext(), _C.RasterContext()
    key = ("ctx-test",)
    slot = torch.zeros(16, dtype=torch.int32)
    slot[0], slot[2], slot[14] = 1000, 77, 5
    with a:
        assert _C.current() is a
        assert _C.check_slots([(slot, None, 2000, key)]) is True          # -> a's hints
        with b:                                                           # nests; b's deferred mode is b's alone
            assert _C.current() is b
            with _C.deferred_capacity_check():
                assert b.deferred and not a.deferred and not _C._default.deferred
            with _C.gradient_buffers(dL_dopacity=torch.zeros(4, 1)):
                assert "dL_dopacity" in b.grad_out and not a.grad_out
            assert not b.grad_out and not b.deferred
            with _C.debug_flags(3):
                sched = _lib.sched_pair(_C.PAIR_K) if not _C.XCD_BLOCK else _lib.sched_xcd_block(_C.XCD_BLOCK)   # (the schedule bits ride along)
                assert b.flags() == 3 | sched and a.flags() == int(_C.DEBUG_FLAGS) | sched
                b.pair_k = 0
                assert b.flags() == 3 | (sched if _C.XCD_BLOCK else 0) and a.flags() == int(_C.DEBUG_FLAGS) | sched
                b.pair_k = None
        assert _C.current() is a
    assert _C.current() is _C._default
    assert a.capacity_hint[key] >= 1000 and a.len_hint[key] == 77 and a.long_tiles_hint[key] == 5
    assert key not in b.capacity_hint and key not in _C._capacity_hint and key not in _C._len_hint
    assert _C.check_slots([(slot, None, 500, key)], context=b) is False and key in b.capacity_hint   # explicit context

    # another thread's default context is its own (the main thread's is the module-level one)
    seen = {}

    def worker():
        seen["ctx"] = _C.current()
        with _C.deferred_capacity_check():
            seen["deferred_here"], seen["deferred_main"] = _C.current().deferred, _C._default.deferred
        with a:
            seen["under_a"] = _C.current() is a
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["ctx"] is not _C._default and seen["deferred_here"] and not seen["deferred_main"] and seen["under_a"]


def test_a_run_without_rgb_loss_only_does_not_silently_train_the_reduced_objective(capsys):
    """The reference's default is --rgb_loss_only=False (lab4d/config.py:161): cycle, feature / reprojection and flow losses are
    then evaluated too (lab4d/engine/trainer.py:477-483 drops them only under the flag).  This build implements the flag's
    objective only: without the flag the entry point stops and names what would be missing (VERDICT r5 missing 2); with
    --allow_rgb_loss_only_semantics it goes on and says so."""
    from vidu4d_amd.lab4d import train
    opts, _ = train.parse_flags(["--fg_motion", "gs-bob"])
    assert opts["rgb_loss_only"] is False
    with pytest.raises(SystemExit) as e:
        train.check_loss_flags(opts)
    for word in ("cycle loss", "feature matching", "optical flow", "--rgb_loss_only", "--allow_rgb_loss_only_semantics"):
        assert word in str(e.value), word
    with pytest.raises(SystemExit):
        train.main(["--fg_motion", "gs-bob"])       # (before anything touches a GPU)
    opts, _ = train.parse_flags(["--allow_rgb_loss_only_semantics"])
    train.check_loss_flags(opts)
    assert "WARNING: --rgb_loss_only is OFF" in capsys.readouterr().out
    opts, _ = train.parse_flags(["--rgb_loss_only"])
    train.check_loss_flags(opts)
    assert capsys.readouterr().out == ""


def test_replayed_counters_carry_their_provenance():
    """bench.py's `roofline.traffic` / `roofline.limiter` are not measured by the run that prints them: they are replayed from
    profiles/pmc_traffic.json (rocprofv3 counter passes need runs of their own) and the line says so in `roofline.replayed_from`
    -- which needs the file to carry the tree and the day it was measured on (tools/stamp_profiles.py)."""
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    meta = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json"))).get("_meta")
    assert meta and re.fullmatch(r"[0-9a-f]{7,40}", meta["commit"]) and re.fullmatch(r"\d{4}-\d{2}-\d{2}", meta["date"]), meta
    src = open(os.path.join(root, "bench.py")).read()
    assert '"replayed_from": replayed_from' in src and 'meta.get("commit")' in src
