"""The random scenes of the footprint fuzzes (tools/fuzz_footprint_cpu.py on the host-compiled header,
tools/fuzz_footprint_gpu.py and tests/test_gpu_round4.py on the device): same generator, same seeds."""
import numpy as np
import torch

from vidu4d_amd.synthetic import make_object_scene, make_scene


def random_scene(rng, large=False):
    """One fuzz scene: size, footprint scale, orientation, near-plane fraction and opacity drawn from wide ranges.
    large: image sizes up to 1920 x 1080 (pixel coordinates in the thousands), fewer surfels."""
    if large:
        W, H = [(512, 384), (1024, 512), (1920, 1080)][int(rng.integers(3))]
        N = int(rng.choice([100, 250]))
    else:
        W, H = int(rng.choice([48, 80, 112, 160])), int(rng.choice([48, 64, 96]))
        N = int(rng.choice([300, 800, 2000]))
    sp = float(rng.choice([0.15, 0.7, 1.5, 4.0, 12.0, 40.0]))
    seed = int(rng.integers(1 << 30))
    sc = (make_object_scene(N, W, H, radius=float(rng.choice([0.2, 0.6])), seed=seed, sigma_px=sp) if rng.random() < 0.4
          else make_scene(N, W, H, seed=seed, sigma_px=sp, big_fraction=float(rng.choice([0.0, 0.2]))))
    g = torch.Generator().manual_seed(seed)
    if rng.random() < 0.5:      # random orientations, elongated footprints
        q = torch.randn(sc.rotations.shape, generator=g)
        sc.rotations = (q / q.norm(dim=1, keepdim=True)).contiguous()
        sc.scales[::2, int(rng.integers(2))] *= float(rng.choice([1e-3, 0.1, 0.3]))
    if rng.random() < 0.4:      # close to the near plane: strong perspective inside one footprint
        k = int(rng.integers(2, 6))
        sc.means3D[::k, 2] = 0.21 + 0.6 * torch.rand(sc.means3D[::k].shape[0], generator=g)
    if rng.random() < 0.3:
        sc.opacities[:] = float(rng.choice([0.004, 0.05, 0.99]))
    return sc, f"{W}x{H} N={N} sigma={sp} seed={seed}"
