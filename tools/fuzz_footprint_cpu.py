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


from tools.fuzz_scenes import random_scene  # noqa: E402  (shared with the GPU fuzz, tools/fuzz_footprint_gpu.py)


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
