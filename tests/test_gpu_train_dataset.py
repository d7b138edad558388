"""lab4d/train.py end to end on the GPU over a sequence in the reference's database/ layout (written by
tests/golden/dataset_fixture.py): config -> VidDataset per video -> data_info (frame offsets, object cameras) ->
DeformableSurfels -> Stage-3 steps -> checkpoint + PLY in the reference's names."""
import importlib.util
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.gpu
def test_train_main_reads_the_reference_layout(tmp_path, gpu_device, capsys):
    spec = importlib.util.spec_from_file_location("dataset_fixture", os.path.join(G, "dataset_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    fx.write_dataset(str(tmp_path), seed=0)
    from vidu4d_amd.lab4d import train
    logroot = str(tmp_path / "logdir")
    train.main(["--seqname", fx.SEQ, "--logname", "t", "--logroot", logroot, "--fg_motion", "gs-bob",
                "--data_root", str(tmp_path / "database"), "--data_prefix", "crop", "--train_res", "16",
                "--feature_type", fx.FEATURE_TYPE, "--delta_list", "2,4", "--num_rounds", "1", "--iters_per_round", "3",
                "--num_surfels", "2000", "--gs_optim_warp=False", "--allow_random_warp", "--save_freq", "1", "--rgb_loss_only"])
    out = capsys.readouterr().out
    assert "2 video(s), 12 frames (crop-16)" in out and "round 0: 3 steps" in out
    run = os.path.join(logroot, f"{fx.SEQ}-t")
    assert os.path.exists(os.path.join(run, "ckpt_latest.pth"))
    ck = torch.load(os.path.join(run, "ckpt_latest.pth"), map_location="cpu", weights_only=False)
    # per-video tables follow the dataset: 2 videos
    keys = [k for k in ck["model"] if k.endswith("camera_mlp.base_quat")]
    assert keys and ck["model"][keys[0]].shape == (2, 4)


@pytest.mark.gpu
def test_train_main_with_the_reference_default_of_networks_that_train(tmp_path, gpu_device, capsys):
    """The same entry with --gs_optim_warp left at the reference's default (True, lab4d/config.py:157) and AdamW starting
    inside the run: the fused warp with captured network graphs is what train.py runs, the networks' parameters move, the
    checkpoint carries them."""
    spec = importlib.util.spec_from_file_location("dataset_fixture", os.path.join(G, "dataset_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    fx.write_dataset(str(tmp_path), seed=0)
    from vidu4d_amd.lab4d import train
    logroot = str(tmp_path / "logdir")
    common = ["--seqname", fx.SEQ, "--logroot", logroot, "--fg_motion", "gs-bob", "--data_root", str(tmp_path / "database"),
              "--data_prefix", "crop", "--train_res", "16", "--feature_type", fx.FEATURE_TYPE, "--delta_list", "2,4",
              "--num_surfels", "2000", "--allow_random_warp", "--save_freq", "1", "--rgb_loss_only"]
    train.main(common + ["--logname", "frozen", "--num_rounds", "1", "--iters_per_round", "1", "--gs_optim_warp=False"])
    train.main(common + ["--logname", "train", "--num_rounds", "2", "--iters_per_round", "4", "--optim_warp_neus_iters", "3"])
    out = capsys.readouterr().out
    assert "round 1: 4 steps" in out
    a = torch.load(os.path.join(logroot, f"{fx.SEQ}-frozen", "ckpt_latest.pth"), map_location="cpu", weights_only=False)["model"]
    b = torch.load(os.path.join(logroot, f"{fx.SEQ}-train", "ckpt_latest.pth"), map_location="cpu", weights_only=False)["model"]
    nets = [k for k in a if (".warp." in k or ".camera_mlp." in k) and a[k].dtype.is_floating_point and "scale" not in k.split(".")[-1]]
    moved = [k for k in nets if k in b and a[k].shape == b[k].shape and not torch.equal(a[k], b[k])]
    assert len(moved) >= 20, (len(moved), len(nets))
    assert all(torch.isfinite(b[k]).all() for k in nets if k in b)
