import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import diff_surfel_rasterization as dsr
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import frame_motion, make_object_scene
dev = torch.device("cuda:0")
_C._SPLIT = "0"
sc = make_object_scene(30_000, 192, radius=0.7, opacity_mode="init").to(dev)
frames = [frame_motion(sc, f, 16) for f in (2, 9)]
rs = dsr.GaussianRasterizationSettings(sc.height, sc.width, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                       sc.sh_degree, sc.campos, False, False)
m = torch.stack([f.means3D for f in frames]); r = torch.stack([f.rotations for f in frames])
for k in (0, 15):
    with _C.debug_flags(_lib.sched_pair(k)), torch.no_grad():
        color, radii, others = dsr.rasterize_frames(m, torch.zeros_like(m), sc.shs, sc.opacities, sc.scales, r, [rs, rs])[:3]
        for i, f in enumerate(frames):
            c1, r1, o1 = dsr.GaussianRasterizer(rs)(means3D=f.means3D, means2D=torch.zeros_like(f.means3D), shs=sc.shs,
                                                     opacities=sc.opacities, scales=sc.scales, rotations=f.rotations)[:3]
            print("K", k, "frame", i, "color diff", float((color[:, i] - c1).abs().max()), int((color[:, i] != c1).sum()),
                  "others", [float((others[p, i] - o1[p]).abs().max()) for p in range(8)], "radii", bool(torch.equal(radii[i], r1)))
