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
