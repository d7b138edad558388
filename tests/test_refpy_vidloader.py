"""vidu4d_amd.lab4d.vidloader against the IMPORTED reference loader (tests/golden/refpy_vidloader.npz, written by
make_refpy_golden.py::gen_vidloader from lab4d/dataloader/vidloader.py + data_utils.py reading the same seeded
directory): every modality (rgb, mask, vis2d, depth, flow fw/bw at several deltas, features incl. the bilinear
pixel sampling, crop2raw, is_detected, cameras), the pair / delta / pixel sampling and the dataset metadata."""
import configparser
import importlib.util
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _fixture():
    spec = importlib.util.spec_from_file_location("dataset_fixture", os.path.join(G, "dataset_fixture.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    return fx


def test_every_modality_and_metadata_match_the_reference_loader(tmp_path):
    from vidu4d_amd.lab4d import vidloader as vl
    fx = _fixture()
    cfg = fx.write_dataset(str(tmp_path), seed=0)
    datasets = vl.config_to_datasets(dict(fx.OPTS), config_path=cfg)
    assert len(datasets) == len(fx.VIDEOS)
    got = fx.collect(datasets, vl.get_data_info)
    want = np.load(os.path.join(G, "refpy_vidloader.npz"))
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = got[k], want[k]
        assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
        assert np.array_equal(a, b), k


def test_stage3_batch_from_the_dataset(tmp_path):
    """frameid = raw frame id + the video's raw offset (model.py:385-399); Kinv = K2inv(ks) @ K2mat(crop2raw)
    (model.py:401-427), on the host."""
    from vidu4d_amd.lab4d import vidloader as vl
    fx = _fixture()
    cfg = fx.write_dataset(str(tmp_path), seed=0)
    datasets = vl.config_to_datasets(dict(fx.OPTS), config_path=cfg)
    info = vl.get_data_info(datasets)
    b = vl.stage3_batch(datasets, info, [(0, 3), (1, 2)])
    assert b["frameid"].tolist() == [4, 9 + 2]  # video 0 kept frame 3 is raw frame 4; video 1 starts at raw offset 9
    assert b["rgb"].shape == (2, 16, 16, 3) and b["mask"].shape == (2, 16, 16, 1) and b["depth"].shape == (2, 16, 16, 1)
    assert b["Kinv"].device.type == "cpu"
    ks, c2r = torch.tensor(fx.KS[1]), torch.as_tensor(datasets[1].crop2raw[2])
    px = torch.tensor([3.0, 5.0, 1.0])
    raw = torch.stack([c2r[0] * px[0] + c2r[2], c2r[1] * px[1] + c2r[3]])
    ray = torch.stack([(raw[0] - ks[2]) / ks[0], (raw[1] - ks[3]) / ks[1], torch.tensor(1.0)])
    assert torch.allclose(b["Kinv"][1] @ px, ray, atol=1e-6)
