#!/usr/bin/env python
"""Recorded segments vs the whole-tile backward vs the CPU oracle: largest gradient difference over the tensor's scale
(MEASUREMENT / test infrastructure; GPU box).  -> profiles/r05_recorded_precision.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import surfel_oracle as so  # noqa: E402
from tests.test_gpu_round4 import GRAD_NAMES, _run  # noqa: E402
from tests.util import oracle_forward  # noqa: E402
from vidu4d_amd import _C, _lib  # noqa: E402
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads  # noqa: E402

dev = torch.device("cuda:0")
_C._SPLIT = "0"
so.set_threads(min(64, os.cpu_count() or 1))
CASES = (("headline 200k 512^2", make_scene(200_000, 512, 512), 1.0),
         ("uniform 60k 256^2", make_scene(60_000, 256, 256, seed=31), 1.0),
         ("object ball r=0.5, init opacity", make_object_scene(40_000, 256, radius=0.5, opacity_mode="init"), 1.0),
         ("uniform 60k, distortion gradient x100", make_scene(60_000, 256, 256, seed=31), 100.0),
         ("thin object r=0.3, distortion gradient x100", make_object_scene(40_000, 256, radius=0.3, opacity_mode="init"), 100.0))
for name, sc, boost in CASES:
    dc, do = make_upstream_grads(sc.width, sc.height)
    do = do.clone()
    do[6] *= boost
    st = oracle_forward(sc)
    g = so.backward(st, dc, do)
    dcd, dod = dc.to(dev), do.to(dev)
    w1 = _run(sc, dev, dcd, dod, flags=_lib.DEBUG_WHOLE_TILE_BACKWARD)
    w2 = _run(sc, dev, dcd, dod, flags=_lib.DEBUG_WHOLE_TILE_BACKWARD)
    r1 = _run(sc, dev, dcd, dod, flags=0)

    def vs_oracle(a):
        """worst error over scale, and how many entries (of which tensor) are beyond 1e-5 -- a threshold flip (T > 0.5,
        T < 1e-4 within an ulp) moves the rows of two surfels by a whole sample, which the x100 cases magnify"""
        worst, where = 0.0, ""
        for k in GRAD_NAMES:
            if k in g:
                e = np.abs(a["grads"][k].cpu().numpy() - g[k]) / (np.abs(g[k]).max() + 1e-30)
                if e.max() > worst:
                    worst, where = float(e.max()), f"{k}, {int((e > 1e-5).sum())} of {e.size} entries beyond 1e-5"
        nc = a["n_contrib"].numpy().view(np.uint32).reshape(2, sc.height, sc.width)
        flips = int(((nc[0] != st["n_contrib"][0]) | (nc[1] != st["n_contrib"][1])).sum())
        return f"{worst:.1e} ({where}; {flips} pixels with a threshold flip)" if worst > 1e-5 else f"{worst:.1e}"

    def vs(a, b):
        return max(float((a["grads"][k] - b["grads"][k]).abs().max()) / (float(b["grads"][k].abs().max()) + 1e-30) for k in GRAD_NAMES)

    print(f"{name}: whole-tile twice (atomics' noise) {vs(w1, w2):.1e} | recorded vs whole-tile {vs(r1, w1):.1e} | "
          f"whole-tile vs oracle {vs_oracle(w1)} | recorded vs oracle {vs_oracle(r1)}", flush=True)
