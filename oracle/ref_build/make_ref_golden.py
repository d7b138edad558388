"""Runs the reference (oracle/_ref/libref_surfel.so) on an MI355X and writes golden fixtures
produced BY THE REFERENCE ITSELF to gpurun_out/ref_golden/ref_*.npz; they are then committed under
tests/golden/.  Usage on the GPU box:  python oracle/ref_build/make_ref_golden.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.ref_build import ref  # noqa: E402
from vidu4d_amd.synthetic import make_scene, make_upstream_grads  # noqa: E402

# name -> (reference build variant, scene).  "strict" = fp contraction off + exact rsqrt (build_ref.py): the
# reference's arithmetic in source order, which the oracle reproduces bit for bit on every binning integer.
STRICT_CASES = {
    "strict_mid": dict(n=6000, width=128, height=128, seed=31),
}
CASES = {
    "tiny": dict(n=64, width=32, height=32, seed=5),
    "ragged": dict(n=600, width=70, height=50, seed=7, bg=(0.2, 0.5, 0.7)),
    "subpixel_deg2": dict(n=500, width=48, height=48, seed=19, sigma_px=0.15, sh_degree=2),
    "huge": dict(n=120, width=64, height=48, seed=23, sigma_px=20.0, big_fraction=0.1),
}

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out_dir = os.path.join(ROOT, "gpurun_out", "ref_golden")
    os.makedirs(out_dir, exist_ok=True)
    jobs = [(n, kw, "default") for n, kw in CASES.items()] + [(n, kw, "strict") for n, kw in STRICT_CASES.items()]
    for name, kw, variant in jobs:
        if not ref.available(variant):
            print("skipping", name, "(library variant not built)")
            continue
        ref.use(variant)
        sc = make_scene(**kw)
        d = sc.to(dev)
        dc, do = make_upstream_grads(sc.width, sc.height)
        rf = ref.forward(d)
        rg = ref.backward(d, rf, dc.to(dev), do.to(dev))
        R = rf["num_rendered"]
        gx, gy = (sc.width + 15) // 16, (sc.height + 15) // 16
        out = dict(means3D=sc.means3D.numpy(), opacities=sc.opacities.numpy(), scales=sc.scales.numpy(),
                   rotations=sc.rotations.numpy(), shs=sc.shs.numpy(), viewmatrix=sc.viewmatrix.numpy(),
                   projmatrix=sc.projmatrix.numpy(), campos=sc.campos.numpy(), bg=sc.bg.numpy(), W=sc.width,
                   H=sc.height, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, sh_degree=sc.sh_degree, dL_dcolor=dc.numpy(),
                   dL_dothers=do.numpy(), radii=rf["radii"].cpu().numpy(), color=rf["color"].cpu().numpy(),
                   others=rf["others"].cpu().numpy(), point_list=ref.state("point_list", R),
                   ranges=ref.state("ranges", gx * gy * 2).reshape(-1, 2),
                   n_contrib=ref.state("n_contrib", 2 * sc.width * sc.height).reshape(2, sc.height, sc.width),
                   tiles_touched=ref.state("tiles_touched", sc.num_surfels), sorted_keys=ref.state("sorted_keys", R),
                   variant=variant)
        for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"):
            out[k] = rg[k].cpu().numpy()
        path = os.path.join(out_dir, f"ref_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))
    ref.use("default")
    # reference timing on this GPU (fwd+bwd, 200k / 512^2), for DESIGN.md / BASELINE.md
    sc = make_scene(200_000, 512).to(dev)
    dc, do = (t.to(dev) for t in make_upstream_grads(512, 512))
    for _ in range(3):
        rf = ref.forward(sc)
        ref.backward(sc, rf, dc, do)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        rf = ref.forward(sc)
        ref.backward(sc, rf, dc, do)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"REFERENCE_ON_MI355X fwd+bwd 200k/512^2: {dt*1e3:.3f} ms/image = {1.0/dt:.1f} images/s "
          f"(reference .cu sources compiled by hipcc, incl. its host sync and allocation-free wrapper)")
    open(os.path.join(out_dir, "reference_timing.txt"), "w").write(f"{dt*1e3:.4f} ms/image {1.0/dt:.2f} images/s\n")
