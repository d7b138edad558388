"""CPU fuzz of the blend culls' footprint test (no GPU): the product's arithmetic header compiled for the host
(tests/host_emul) renders random scenes with the test applied PER PIXEL and without it; images, contributor counts and
gradient accumulators must be bit-identical, and the quadrant form must never drop a contributing quadrant.
Usage: python tools/fuzz_footprint_cpu.py [scenes] [seed] [large]"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.host_emul import emul
from tests.util import oracle_forward
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads


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


def check_scene(sc):
    """-> (bit-identical with / without the per-pixel test, quadrant scan counts)"""
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(sc.width, sc.height)
    a = emul.run(st, dc.numpy(), do.numpy(), cull=True)
    b = emul.run(st, dc.numpy(), do.numpy(), cull=False)
    same = all(np.array_equal(a[k], b[k]) for k in ("color", "others", "n_contrib", "final_T", "acc"))
    return same, emul.footprint_scan(st)


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    tot = dict(kept=0, contributing=0, dropped_contributing=0, box_would_keep=0)
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    for i in range(n_scenes):
        sc, what = random_scene(rng, large)
        same, c = check_scene(sc)
        for k in tot:
            tot[k] += c[k]
        if not same or c["dropped_contributing"]:
            bad += 1
            print(f"MISMATCH scene {i}: {what} identical={same} scan={c}", flush=True)
    print(f"{n_scenes} scenes, {bad} mismatching; (surfel, quadrant) pairs: kept {tot['kept']}, contributing {tot['contributing']}, "
          f"dropped although contributing {tot['dropped_contributing']}, a bounding box would keep {tot['box_would_keep']}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
