"""hipGraph replay of the bare rasterizer forward + backward on inputs that change between replays,
compared with eager execution of the same calls (reproducer for the graph-replay problem)."""
import faulthandler; faulthandler.enable()
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd import _C
from vidu4d_amd.synthetic import make_upstream_grads, make_object_scene
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "200000")); W = H = 512
sc = make_object_scene(N, W, H, radius=1.0).to(dev)
dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
empty = torch.empty(0, device=dev)
means = sc.means3D.clone()
def run():
    out = _C.rasterize_gaussians(sc.bg, means, empty, sc.opacities, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix, sc.projmatrix,
                                 sc.tanfovx, sc.tanfovy, H, W, sc.shs, 3, sc.campos, False, False)
    R, color, others, radii, geom, binning, img = out
    g = _C.rasterize_gaussians_backward(sc.bg, means, radii, empty, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix, sc.projmatrix,
                                        sc.tanfovx, sc.tanfovy, dc, do, sc.shs, 3, sc.campos, geom, R, binning, img, False)
    return [color] + [t for t in g if t.numel()]
run()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        with _C.deferred_capacity_check(): run()
        _C.check_deferred()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    with _C.graph_capture_mode() as cap:
        outs = run()
base = means.clone()
names = ["color", "m2D", "colors", "opac", "m3D", "tmat", "sh", "scales", "rot"]
worst = 0.0
for i in range(6):
    ang = 0.05 * i
    means[:, 0] = float(np.cos(ang)) * base[:, 0] - float(np.sin(ang)) * base[:, 1]
    means[:, 1] = float(np.sin(ang)) * base[:, 0] + float(np.cos(ang)) * base[:, 1]
    g.replay(); torch.cuda.synchronize()
    got = [t.clone() for t in outs]
    ref = run(); torch.cuda.synchronize()
    err = {n: float((a - b).abs().max() / max(1e-12, float(b.abs().max()))) for n, a, b in zip(names, got, ref) if float(b.abs().max()) > 0}
    worst = max(worst, max(err.values()))
    print(i, {k: f"{v:.1e}" for k, v in err.items()}, flush=True)
print("REPLAY_OK" if worst < 1e-3 else "REPLAY_BAD", worst)
