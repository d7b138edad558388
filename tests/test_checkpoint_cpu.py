"""Checkpoints in the reference's key layout: a Stage-2 style checkpoint assembled from state dicts that the
IMPORTED reference modules saved (tests/golden/refpy_nets.pt) populates the Stage-3 field's bones and cameras;
a checkpoint whose networks do not fit is refused when they are frozen."""
import os

import numpy as np
import pytest
import torch

from vidu4d_amd.lab4d import checkpoint
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.nets import make_frame_info
from vidu4d_amd.lab4d.stage3 import Stage3Trainer

G = os.path.join(os.path.dirname(__file__), "golden")


def _model(offsets, n=50):
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    m = DeformableSurfels(dict(fg_motion="gs-bob"), num_frames=int(offsets[-1]), device="cpu",
                          data_info={"frame_info": make_frame_info(offsets)})
    m.init_from_points(rng.normal(size=(n, 3)).astype(np.float32) * 0.1, rng.uniform(size=(n, 3)).astype(np.float32))
    return m


def _reference_style_checkpoint(tmp_path, n=70):
    nets = torch.load(os.path.join(G, "refpy_nets.pt"), weights_only=False)["v2"]
    g = torch.Generator().manual_seed(3)
    state = {"module.fields.field_params.fg.warp." + k: v for k, v in nets["warp"].items()}     # DDP prefix as upstream saves
    state.update({"module.fields.field_params.fg.camera_mlp." + k: v for k, v in nets["camera_mlp"].items()})
    surf = {"_xyz": torch.randn(n, 3, generator=g), "_features_dc": torch.randn(n, 1, 3, generator=g),
            "_features_rest": torch.randn(n, 15, 3, generator=g), "_opacity": torch.randn(n, 1, generator=g),
            "_scaling": torch.randn(n, 2, generator=g), "_rotation": torch.randn(n, 4, generator=g),
            "_regist_feat": torch.randn(n, 16, generator=g), "logsigma": torch.zeros(1), "logibeta": torch.ones(1),
            "aabb": torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), "learnable_bkgd": torch.tensor([0.1, 0.2, 0.3])}
    state.update({"module.fields.field_params.fg." + k: v for k, v in surf.items()})
    state["module.intrinsics.base_logfocal"] = torch.zeros(2)      # a key of another sub-module: ignored
    path = os.path.join(tmp_path, "ckpt_0020.pth")
    torch.save({"current_steps": 4000, "current_round": 20, "model": state, "optimizer": {}}, path)
    return path, nets, surf


def test_stage2_style_checkpoint_populates_networks(tmp_path):
    path, nets, surf = _reference_style_checkpoint(str(tmp_path))
    m = _model(nets["offsets"])
    tr = Stage3Trainer(m, dict(fg_motion="gs-bob", gs_optim_warp=False))
    info = checkpoint.load_checkpoint(path, m, tr, map_location="cpu")
    assert info["network_keys_not_loaded"] == []
    assert not [k for k in info["missing_keys"] if k.startswith(("warp.", "camera_mlp."))]
    for k, v in nets["warp"].items():
        assert torch.equal(m.warp.state_dict()[k], v), k
    for k, v in nets["camera_mlp"].items():
        assert torch.equal(m.camera_mlp.state_dict()[k], v), k
    assert m._xyz.shape[0] == surf["_xyz"].shape[0] and torch.equal(m._xyz.detach(), surf["_xyz"])
    assert torch.equal(m.learnable_bkgd.detach(), surf["learnable_bkgd"]) and torch.equal(m.aabb, surf["aabb"])
    assert tr.current_steps == 0 and tr.gs_optimizer.param_groups[0]["params"][0] is m._xyz   # reset_steps, fresh optimizer
    assert all(not p.requires_grad for p in m.warp.parameters())


def test_checkpoint_with_other_frame_layout_is_refused_when_networks_are_frozen(tmp_path):
    path, nets, _ = _reference_style_checkpoint(str(tmp_path))
    m = _model([0, 40])                    # one video instead of two: instance tables have other shapes
    tr = Stage3Trainer(m, dict(fg_motion="gs-bob", gs_optim_warp=False))
    with pytest.raises(RuntimeError, match="randomly\\s+initialised bones"):
        checkpoint.load_checkpoint(path, m, tr, map_location="cpu")
    # trainable networks (or an explicit waiver) may start from what does fit
    m2 = _model([0, 40])
    tr2 = Stage3Trainer(m2, dict(fg_motion="gs-bob", gs_optim_warp=True, num_rounds=2, iters_per_round=10))
    info = checkpoint.load_checkpoint(path, m2, tr2, map_location="cpu")
    assert info["network_keys_not_loaded"]
    m3 = _model([0, 40])
    tr3 = Stage3Trainer(m3, dict(fg_motion="gs-bob", gs_optim_warp=False))
    checkpoint.load_checkpoint(path, m3, tr3, map_location="cpu", allow_random_networks=True)


def test_save_load_round_trip(tmp_path):
    m = _model([0, 12])
    tr = Stage3Trainer(m, dict(fg_motion="gs-bob"))
    tr.current_steps = 77
    p = checkpoint.save_checkpoint(tr, str(tmp_path), round_count=0)
    assert p and os.path.exists(os.path.join(str(tmp_path), "ckpt_latest.pth")) and os.path.exists(
        os.path.join(str(tmp_path), "000-fg-gs.ply"))
    m2 = _model([0, 12], n=20)
    tr2 = Stage3Trainer(m2, dict(fg_motion="gs-bob"))
    info = checkpoint.load_checkpoint(p, m2, tr2, map_location="cpu", reset_steps=False)
    assert tr2.current_steps == 77 and info["network_keys_not_loaded"] == []
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k
