"""Forward+backward time of two stacked frames per blend instance on the same dense object scene: what the backward of the
planes-0-4 instance costs against the colour + alpha one and the full one (same lists, same walks)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import diff_surfel_rasterization as dsr
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import frame_motion, make_object_scene, make_upstream_grads
dev = torch.device("cuda:0")
_C._SPLIT = "0"
N, W = 200_000, 512
sc = make_object_scene(N, W, None, radius=1.0, seed=1234, opacity_mode="init").to(dev)
frames = [frame_motion(sc, f, 120) for f in range(4)]
rs = dsr.GaussianRasterizationSettings(sc.height, sc.width, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                       sc.sh_degree, sc.campos, False, False)
dc, do = (t.to(dev) for t in make_upstream_grads(W, sc.height))
for mode, aux, keep in (("full", 0, list(range(8))), ("alpha", _lib.AUX_ALPHA, [1]), ("geom", _lib.AUX_GEOM, [0, 1, 2, 3, 4])):
    z = torch.zeros_like(do); z[keep] = do[keep]
    dcs, dos = torch.stack([dc, dc], 1).contiguous(), torch.stack([z, z], 1).contiguous()
    leaves = [t.clone().requires_grad_(True) for t in (sc.shs, sc.opacities, sc.scales)]
    def step(i, backward):
        ids = [(2 * i) % 4, (2 * i + 1) % 4]
        m = torch.stack([frames[j].means3D for j in ids]).requires_grad_(backward)
        r = torch.stack([frames[j].rotations for j in ids]).requires_grad_(backward)
        with torch.set_grad_enabled(backward):
            color, radii, allmap = dsr.rasterize_frames(m, torch.zeros_like(m, requires_grad=backward), leaves[0], leaves[1], leaves[2], r, [rs, rs], aux_planes=aux)
            if backward:
                torch.autograd.backward([color, allmap], [dcs, dos])
                for l in leaves: l.grad = None
    res = []
    for backward in (False, True):
        for i in range(8): step(i, backward)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(30): step(i, backward)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e6)
        res.append(best)
    print(mode, "forward", round(res[0], 1), "forward+backward", round(res[1], 1), "backward side", round(res[1] - res[0], 1), "us per two frames", flush=True)
