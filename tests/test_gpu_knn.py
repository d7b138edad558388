"""GPU parity of the 3-NN scale initialiser (csrc/knn.hip, replaces simple-knn's distCUDA2,
simple_knn.cu:185-221) against an exact k-d tree query; fp32 tolerance 1e-5 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _want(pts):
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(axis=1)


@pytest.mark.parametrize("n", [4, 255, 256, 257, 5000, 60_000])
def test_dist2_matches_kdtree(gpu_device, n):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    if n >= 255:
        pts[7] = pts[3]          # coincident points count with distance 0 (excluded by index, not by value)
        pts[n // 2] *= 40.0      # an outlier far from everything
    got = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
    want = _want(pts)
    assert got.shape == (n,) and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-5 * want.max() + 1e-9
    rel = np.abs(got - want) / np.maximum(want, 1e-12)
    assert np.median(rel) < 1e-6


def test_dist2_edge_cases(gpu_device):
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.empty(0, 3, device=gpu_device)).shape == (0,)
    few = distCUDA2(torch.rand(3, 3, device=gpu_device))  # fewer than 4 points: FLT_MAX terms, as upstream
    assert bool((few > 1e37).all())
    with pytest.raises(RuntimeError):
        distCUDA2(torch.rand(5, 3))
    with pytest.raises(RuntimeError):
        distCUDA2(torch.rand(5, 2, device=gpu_device))


def test_scale_init_uses_the_kernel(gpu_device):
    from vidu4d_amd.gs.gaussian_model import GaussianModel, mean_knn_dist2
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(3000, 3)).astype(np.float32)

    class P:
        points, colors = pts, rng.uniform(size=(3000, 3)).astype(np.float32)
    m = GaussianModel(3, device=gpu_device)
    m.create_from_pcd(P, 1.0)
    want = np.log(np.sqrt(np.maximum(mean_knn_dist2(pts), 1e-7)))
    assert np.abs(m._scaling.detach().cpu().numpy()[:, 0] - want).max() < 1e-4


def test_radius_neighbor_count_matches_kdtree_and_prunes_like_open3d(gpu_device):
    """The neighbour count behind the Stage-3 outlier pass (open3d remove_radius_outlier(nb_points=20,
    radius=0.004), trainer.py:573-588): count includes the point itself, a point is kept when count > 20."""
    from scipy.spatial import cKDTree
    from vidu4d_amd.simple_knn import radius_neighbor_count
    rng = np.random.default_rng(5)
    dense = rng.normal(size=(6000, 3)).astype(np.float32) * 0.01     # a blob: many neighbours within 0.004
    stray = rng.uniform(-1, 1, size=(300, 3)).astype(np.float32)      # isolated points
    pts = np.concatenate([dense, stray])
    r = 0.004
    got = radius_neighbor_count(torch.from_numpy(pts).to(gpu_device), r).cpu().numpy()
    tree = cKDTree(pts.astype(np.float64))
    want = np.array([len(ix) for ix in tree.query_ball_point(pts.astype(np.float64), r)])
    # (pairs within fp32 rounding of the radius may fall on either side)
    d = np.abs(got - want)
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    keep = got > 20
    assert keep[:6000].mean() > 0.5 and not keep[6000:].any()
    assert radius_neighbor_count(torch.empty(0, 3, device=gpu_device), r).shape == (0,)


def test_trainer_runs_the_outlier_pass(gpu_device):
    from tests.test_gpu_stage3 import _model
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device
    m = _model(dev, n=3000, seed=2, densify_from_iter=0, densification_interval=1000, outlier_filtering_interval=3)
    with torch.no_grad():
        m._xyz[:50] += 0.6  # strays: no neighbours within 0.004 ... (neither have most others at this density)
    tr = Stage3Trainer(m)
    tr.current_steps = 1
    n0 = m._xyz.shape[0]
    for _ in range(3):
        tr.train_step(synthetic_batch(m, [0, 1], 64, 64))
    assert m._xyz.shape[0] < n0 and m._xyz.shape[0] == m._opacity.shape[0] == m.max_radii2D.shape[0]
    assert tr.gs_optimizer.state[m._xyz]["exp_avg"].shape[0] == m._xyz.shape[0]
