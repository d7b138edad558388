"""Generates the oracle golden fixtures (tests/golden/oracle_*.npz).

Run from the repo root:  python tests/golden/make_golden.py
Inputs are the seeded synthetic scenes of vidu4d_amd/synthetic.py; outputs come from the C oracle
(oracle/surfel_oracle.c).  The fixtures pin the oracle against accidental edits; fixtures generated
by the reference's own sources compiled for gfx950 (ref_*.npz) are produced on the GPU box by
oracle/ref_build/make_ref_golden.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import surfel_oracle as so  # noqa: E402
from vidu4d_amd.synthetic import make_scene, make_upstream_grads  # noqa: E402

CASES = {
    "tiny": dict(n=64, width=32, height=32, seed=5),
    "ragged": dict(n=600, width=70, height=50, seed=7, bg=(0.2, 0.5, 0.7)),
    "subpixel_deg2": dict(n=500, width=48, height=48, seed=19, sigma_px=0.15, sh_degree=2),
}

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, kw in CASES.items():
        sc = make_scene(**kw)
        st = so.forward(sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.viewmatrix, sc.projmatrix, sc.campos,
                        sc.bg, sc.width, sc.height, sc.tanfovx, sc.tanfovy, sc.sh_degree, shs=sc.shs)
        dc, do = make_upstream_grads(sc.width, sc.height)
        g = so.backward(st, dc, do)
        out = dict(means3D=sc.means3D.numpy(), opacities=sc.opacities.numpy(), scales=sc.scales.numpy(),
                   rotations=sc.rotations.numpy(), shs=sc.shs.numpy(), viewmatrix=sc.viewmatrix.numpy(),
                   projmatrix=sc.projmatrix.numpy(), campos=sc.campos.numpy(), bg=sc.bg.numpy(), W=sc.width,
                   H=sc.height, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, sh_degree=sc.sh_degree, dL_dcolor=dc.numpy(),
                   dL_dothers=do.numpy())
        for k in ("radii", "point_list", "ranges", "n_contrib", "color", "others"):
            out[k] = st[k]
        for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"):
            out[k] = g[k]
        path = os.path.join(here, f"oracle_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))
