#!/usr/bin/env python
"""GPU fuzz of the footprint cull (TEST INFRASTRUCTURE; needs an MI355X): the generator of tools/fuzz_footprint_cpu.py
run through the PRODUCT kernels.  Per scene:
  (a) cull A/B on the device -- forward planes, final_T, n_contrib must be BIT-IDENTICAL with the culls on and off
      (VIDU4D_DEBUG_NO_CULL: every list entry evaluated for every pixel of its tile, the reference's walk,
      forward.cu:359-405); the sums the blend kernel makes (BLEND_GRADS; float atomics: their order differs run to run)
      within 1e-6 of scale plus four times the run-to-run noise measured on the same scene (culls on, twice), the
      chain-rule tensors behind them (preprocess_bwd, deterministic) within the north star's 1e-4;
  (b) product vs the CPU oracle: radii / n_contrib mismatches and worst error over scale per tensor
      (tests/util.py::assert_close at ORACLE_RTOL = 1e-5 of scale decides; a pixel whose contributor counts differ from
      the oracle's is a threshold flip -- T > 0.5, T < 1e-4 on a transmittance within an ulp -- and moves the gradient rows
      of TWO surfels, the one that loses the sample and the one that gains it: that many rows are allowed per flip).
      The chain-rule tensors (dL_dmeans3D / dL_dscales / dL_drotations; preprocess_bwd) are held to the north star's 1e-4
      against the oracle directly, and to 1e-5 against the oracle's chain applied to the PRODUCT's own blend sums
      (oracle backward_chain on the returned dL_dtransMat / dL_dcolors): the chain multiplies a rounding difference of
      those sums by two orders of magnitude (a 1e-6 perturbation of them moves dL_dscales by 1e-4..1e-2 of its scale on
      these scenes; seed 0's scene 29 -- 300 screen-filling surfels, dL_dscales of scale 6e-6 from dL_dtransMat of scale
      5e-2 -- by half of it), so the direct comparison cannot be tighter than the sums' own 1e-5 times that factor.
Usage: python tools/fuzz_footprint_gpu.py [scenes=200] [seed=0] [large] > profiles/r04_fuzz_footprint_gpu.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import surfel_oracle as so  # noqa: E402
from tests.test_gpu_round4 import BLEND_GRADS, GRAD_NAMES, _grad_error, _run  # noqa: E402
import functools  # noqa: E402

from tests.util import DIST_ATOL, ORACLE_RTOL, RTOL, oracle_forward  # noqa: E402
from tests.util import assert_close as _assert_close  # noqa: E402

assert_close = functools.partial(_assert_close, rtol=ORACLE_RTOL)
from tools.fuzz_scenes import random_scene  # noqa: E402
from vidu4d_amd import _lib  # noqa: E402
from vidu4d_amd.synthetic import make_upstream_grads  # noqa: E402


CHAIN = ("dL_dmeans3D", "dL_dscales", "dL_drotations")


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    so.set_threads(min(64, os.cpu_count() or 1))
    bad_ab = bad_oracle = flips_total = 0
    worst_noise = worst_ab = worst_chain = 0.0
    worst = {}
    pairs = 0
    for i in range(n_scenes):
        sc, what = random_scene(rng, large)
        dc, do = make_upstream_grads(sc.width, sc.height)
        a = _run(sc, dev, dc.to(dev), do.to(dev), flags=0)
        a2 = _run(sc, dev, dc.to(dev), do.to(dev), flags=0)
        b = _run(sc, dev, dc.to(dev), do.to(dev), flags=_lib.DEBUG_NO_CULL)
        pairs += int(a["R"])
        diff = [k for k in ("color", "others", "radii", "n_contrib", "final_T") if not torch.equal(a[k], b[k])]
        noise, gerr = _grad_error(a2, a, BLEND_GRADS), _grad_error(a, b, BLEND_GRADS)
        worst_noise, worst_ab = max(worst_noise, noise), max(worst_ab, gerr)
        worst_chain = max(worst_chain, _grad_error(a, b))
        if diff or gerr > 1e-6 + 4.0 * noise or _grad_error(a, b) > 1e-4:
            bad_ab += 1
            print(f"CULL A/B MISMATCH scene {i}: {what} differing={diff} worst gradient error/scale={gerr:.2e} "
                  f"(run-to-run noise {noise:.2e})", flush=True)
        # ---- product vs oracle
        st = oracle_forward(sc)
        g = so.backward(st, dc, do)
        own = so.backward_chain(st, a["grads"]["dL_dtransMat"], None, g["dL_dnormal"], a["grads"]["dL_dcolors"], after_aabb=True)
        fails = []
        W, H = sc.width, sc.height
        if not np.array_equal(a["radii"].cpu().numpy(), st["radii"]):
            fails.append("radii")
        nc = a["n_contrib"].numpy().view(np.uint32).reshape(2, H, W)
        flipped = (nc[0] != st["n_contrib"][0]) | (nc[1] != st["n_contrib"][1])                # pixels with a flipped threshold
        flips = int(flipped.sum())
        rows = int(np.maximum(nc[0], st["n_contrib"][0])[flipped].sum())   # surfels those pixels blend (an upper bound)
        flips_total += flips
        if flips > max(2, int(2e-5 * W * H)):
            fails.append(f"n_contrib differs at {flips} pixels")

        def cmp(name, got, want, **kw):
            if flips and np.ndim(want) == 2 and name.startswith("dL_"):   # per-surfel tensor: the rows of the surfels it blends
                kw["min_outliers"] = max(2 * flips, rows) * int(np.prod(np.shape(want)[1:]))
            elif flips and name.startswith("dL_dsh"):
                kw["min_outliers"] = max(2 * flips, rows) * int(np.prod(np.shape(want)[1:]))
            elif flips:
                kw["min_outliers"] = flips * (3 if name == "color" else 1)
            if flips and (name.startswith("dL_") or name in ("others5", "others7")):
                kw["outlier_rtol"] = 1.0   # (a flipped median sample replaces the pixel's median depth / max weight outright)
            if name in CHAIN:
                kw["rtol"] = RTOL
            try:
                w = assert_close(name, got, want, **kw)
            except AssertionError as e:
                fails.append(str(e)[:120])
                w = float(np.abs(np.asarray(got.detach().cpu() if hasattr(got, "detach") else got, np.float64) - want).max()
                          / (np.abs(want).max() + 1e-30))
            worst[name] = max(worst.get(name, 0.0), w)

        cmp("color", a["color"], st["color"])
        for p in range(8):
            cmp(f"others{p}", a["others"][p], st["others"][p], atol=DIST_ATOL if p == 6 else 0.0)
        for k in GRAD_NAMES:
            if k in g:
                cmp(k, a["grads"][k], g[k])
        for k in CHAIN:
            cmp(k + " | oracle chain of the product's sums", a["grads"][k], own[k])
        if fails:
            bad_oracle += 1
            print(f"PRODUCT != ORACLE scene {i}: {what}: {fails}", flush=True)
    print(f"{n_scenes} scenes (seed {seed}{', large' if large else ''}), {pairs} (surfel, tile) pairs: "
          f"{bad_ab} cull A/B mismatches (forward planes / final_T / n_contrib bit-identical in the others; gradients: worst "
          f"A/B difference {worst_ab:.2e} of scale, worst run-to-run noise {worst_noise:.2e}, chain-rule tensors {worst_chain:.2e}), "
          f"{bad_oracle} scenes outside 1e-5 of scale vs the oracle ({flips_total} pixels with a threshold flip in all)")
    print("worst error / scale vs the oracle over all scenes:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
    sys.exit(1 if (bad_ab or bad_oracle) else 0)


if __name__ == "__main__":
    main()
