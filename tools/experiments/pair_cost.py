"""Forward-only time of two stacked frames per blend instance, one workgroup per tile against paired workgroups
(VIDU4D_SCHED_PAIR K): uniform scene (throughput regime: what a pair costs) and the dense ball (what it buys)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import diff_surfel_rasterization as dsr
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import frame_motion, make_object_scene, make_scene
dev = torch.device("cuda:0")
_C._SPLIT = "0"
N, W = 200_000, 512
scenes = {"uniform": make_scene(N, W, None, seed=1234).to(dev),
          "ball": make_object_scene(N, W, None, radius=1.0, seed=1234, sigma_px=6.0, opacity_mode="init").to(dev)}
KS = [int(k) for k in os.environ.get("KS", "0 6 15").split()]
for name, sc in scenes.items():
    frames = [frame_motion(sc, f, 120) for f in range(4)]
    rs = dsr.GaussianRasterizationSettings(sc.height, sc.width, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                           sc.sh_degree, sc.campos, False, False)
    for mode, aux in (("full", 0), ("alpha", _lib.AUX_ALPHA), ("geom", _lib.AUX_GEOM)):
        row = []
        for k in KS:
            _C.PAIR_K = k
            def step(i):
                ids = [(2 * i) % 4, (2 * i + 1) % 4]
                m = torch.stack([frames[j].means3D for j in ids]); r = torch.stack([frames[j].rotations for j in ids])
                with torch.no_grad():
                    dsr.rasterize_frames(m, torch.zeros_like(m), sc.shs, sc.opacities, sc.scales, r, [rs, rs], aux_planes=aux)
            for i in range(10): step(i)
            best = 1e9
            for rep in range(6):   # (the smallest of six runs of 40 steps: the box's other tenants show up as outliers)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(40): step(i)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e6)
            row.append(round(best, 1))
        print(name, mode, "forward us per two frames at K =", dict(zip(KS, row)), flush=True)
