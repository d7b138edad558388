"""One-off fuzz of the paired walk (not part of the suite): random scenes of tools/fuzz_scenes.py, every blend instance, the
product's rule and every tile paired, against the one-workgroup walk: contributor counts, final transmittance, median sample
identical; sums within 2e-6 of scale; gradients within 1e-4."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_round4 import _fuzz_scenes, _grad_error, _run
from tests.test_gpu_paired_tiles import _planes_close
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import make_upstream_grads
dev = torch.device("cuda:0")
_C._SPLIT = "0"
_C.PAIR_K = 0
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bad = 0
paired = 0
for seed in range(4):
    for sc, what in _fuzz_scenes(n_scenes // 4, 100 + seed, large=(seed == 3)):
        dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
        for mode, aux, keep in (("full", 0, list(range(8))), ("lite", _lib.AUX_ALPHA, [1]), ("geom", _lib.AUX_GEOM, [0, 1, 2, 3, 4])):
            z = torch.zeros_like(do); z[keep] = do[keep]
            b = _run(sc, dev, dc, z, aux=aux, flags=0)
            for k in (6, 15):
                a = _run(sc, dev, dc, z, aux=aux, flags=_lib.sched_pair(k))
                paired += int(a["header"][17]) > 0
                try:
                    _planes_close(a, b, (what, mode, k))
                    assert _grad_error(a, b) <= 1e-4, ("grads", _grad_error(a, b))
                except AssertionError as e:
                    bad += 1
                    print("MISMATCH", what, mode, k, str(e)[:200], flush=True)
print(f"pair fuzz: {n_scenes} scenes x 3 instances x 2 rules, {paired} launches with pairs, {bad} mismatches")
