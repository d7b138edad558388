# one-off: outlier fractions of the world_kcam configuration (tests/test_gpu_reference.py) for its frozen budgets
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle.ref_build import ref
from tests.util import look_at_view, to_np
from tests.test_gpu_reference import _run_product, _n_contrib_mismatch, GRADS
from vidu4d_amd.synthetic import make_scene, make_upstream_grads
from vidu4d_amd.gs.cameras import KCamera
dev = torch.device("cuda:0")
cam = KCamera(H=288, W=384, left=-0.42, right=0.55, top=0.31, bottom=-0.40, data_device="cpu")
tanx, tany = (float(torch.tan(f.float() * 0.5)) for f in (cam.FoVx, cam.FoVy))
sc = make_scene(60_000, 384, 288, seed=77)
view, eye = look_at_view((0.4, -0.3, -0.5), (0.0, 0.1, 3.0))
sc.viewmatrix, sc.campos = view, eye
sc.projmatrix = (view @ cam.projection_matrix.cpu()).contiguous()
sc.tanfovx, sc.tanfovy = tanx, tany
W, H, P = sc.width, sc.height, sc.num_surfels
d = sc.to(dev)
dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
for variant in ("strict", "default"):
    ref.use(variant)
    rf = ref.forward(d); rg = ref.backward(d, rf, dc, do)
    color, allmap, grads, ints = _run_product(d, dc, do, W, H)
    radii = to_np(rf["radii"])
    print(variant, "R", int(rf["num_rendered"]), "radii flips", float((ints["radii"] != radii).mean()),
          "n_contrib", _n_contrib_mismatch(ref.state("n_contrib", 2 * W * H).reshape(2, H, W), ints["n_contrib"]))
    def stat(name, a, b):
        a, b = to_np(a).astype(np.float64), to_np(b).astype(np.float64)
        sc_ = np.abs(b).max(); err = np.abs(a - b)
        print(f"  {name}: frac {float((err > 1e-4 * sc_).mean()):.3e} worst {float(err.max() / (sc_ + 1e-30)):.3e}")
    stat("color", color, rf["color"])
    for i in range(8): stat(f"others{i}", allmap[i], rf["others"][i])
    for k in GRADS: stat(k, grads[k], rg[k])
