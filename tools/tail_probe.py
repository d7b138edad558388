"""How much of a blend launch is tail?  Stage timers of the stacked rasterizer for F = 1, 2, 4, 8 frames of the headline
scene in ONE launch set: per-frame kernel time falls with F by what the ramp-up / tail of a launch costs.
    python tools/tail_probe.py [surfels] [res]"""
import sys

import torch

sys.path.insert(0, ".")
import diff_surfel_rasterization as dsr  # noqa: E402
from vidu4d_amd import _C, _lib  # noqa: E402
from vidu4d_amd.synthetic import frame_motion, make_scene, make_upstream_grads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
scene = make_scene(N, W, None, seed=1234).to(dev)
H = scene.height
dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
frames = [frame_motion(scene, f, 120) for f in range(16)]
rs = dsr.GaussianRasterizationSettings(H, W, scene.tanfovx, scene.tanfovy, scene.bg, 1.0, scene.viewmatrix, scene.projmatrix,
                                       scene.sh_degree, scene.campos, False, False)
opac = scene.opacities.clone().requires_grad_(True)
scales = scene.scales.clone().requires_grad_(True)
shs = scene.shs.clone().requires_grad_(True)
for F in (1, 2, 4, 8):
    dcs, dos = torch.stack([dc] * F, 1).contiguous(), torch.stack([do] * F, 1).contiguous()

    def step(k):
        ids = [(k * F + i) % 16 for i in range(F)]
        m = torch.stack([frames[i].means3D for i in ids]).requires_grad_(True)
        r = torch.stack([frames[i].rotations for i in ids]).requires_grad_(True)
        m2d = torch.zeros_like(m, requires_grad=True)
        with _C.deferred_capacity_check():
            color, radii, allmap = dsr.rasterize_frames(m, m2d, shs, opac, scales, r, [rs] * F)
            torch.autograd.backward([color, allmap], [dcs, dos])
        _C.check_deferred()
        for t in (opac, scales, shs):
            t.grad = None
    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    _lib.profile_read(reset=True)
    _lib.profile_enable(True)
    for k in range(12):
        step(k)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    p = {k: ms / n for k, (ms, n) in _lib.profile_read(reset=True).items() if n}
    print(f"F={F}: per launch " + " ".join(f"{k}={v * 1e3:.0f}us" for k, v in p.items()) +
          f" | per frame blend_fwd {p['blend_fwd'] * 1e3 / F:.0f} blend_bwd {p['blend_bwd'] * 1e3 / F:.0f} total {sum(p.values()) * 1e3 / F:.0f}")
