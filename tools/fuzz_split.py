"""Fuzz: segment-parallel blend vs single-workgroup blend on random scenes (sizes, resolutions, object
radii, opacities, scales), with recycled allocator blocks poisoned by NaNs.  Prints mismatches."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vidu4d_amd import _C
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
empty = torch.empty(0, device=dev)

def poison():
    blocks = [torch.full((32 * 1024 * 1024,), float("nan"), device=dev) for _ in range(4)]
    del blocks

def run(sc, dc, do, mode):
    _C._SPLIT = mode
    poison()
    out = _C.rasterize_gaussians(sc.bg, sc.means3D, empty, sc.opacities, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                 sc.projmatrix, sc.tanfovx, sc.tanfovy, sc.height, sc.width, sc.shs, 3, sc.campos, False, False)
    R, color, others, radii, geom, binning, img = out
    poison()
    g = _C.rasterize_gaussians_backward(sc.bg, sc.means3D, radii, empty, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                        sc.projmatrix, sc.tanfovx, sc.tanfovy, dc, do, sc.shs, 3, sc.campos, geom, R, binning, img, False)
    ncon = _C.read_state("n_contrib", None, geom, binning, img, sc.num_surfels, sc.width, sc.height, torch.int32, 2 * sc.width * sc.height)
    rg = _C.read_state("ranges", None, geom, binning, img, sc.num_surfels, sc.width, sc.height, torch.int32, 2 * ((sc.width + 15) // 16) * ((sc.height + 15) // 16)).view(-1, 2)
    return R, color, others, ncon, [t for t in g if t.numel()], int((rg[:, 1] - rg[:, 0]).max())

t0 = time.time(); n = 0; bad = 0; nlim = 0; nfit = 0
while time.time() - t0 < budget:
    N = int(rng.choice([3000, 20000, 60000, 150000]))
    W = int(rng.choice([96, 130, 256, 400, 512])); H = int(rng.choice([80, 128, 256, 333, 512]))
    radius = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
    kind = rng.choice(["object", "uniform"])
    om = str(rng.choice(["random", "init"]))
    sp = float(rng.choice([0.7, 1.5, 4.0]))
    sc = (make_object_scene(N, W, H, radius=radius, seed=int(rng.integers(1 << 30)), opacity_mode=om, sigma_px=sp) if kind == "object"
          else make_scene(N, W, H, seed=int(rng.integers(1 << 30)), opacity_mode=om, sigma_px=sp)).to(dev)
    if rng.random() < 0.3:
        sc.opacities[:] = float(rng.choice([0.02, 0.3, 0.9]))
    dc, do = (t.to(dev) for t in make_upstream_grads(W, H, seed=int(rng.integers(1 << 30))))
    a = run(sc, dc, do, "0"); b = run(sc, dc, do, "1")
    n += 1
    msgs = []
    # segment-limited deferred call with the hint set to a fraction of the true depth: either it is flagged
    # as truncated, or it must equal the unlimited result
    depth = int(b[3][: W * H].max())
    if b[5] > 1024 and depth > 2048:
        key = (sc.num_surfels, W, H, str(dev))
        _C._unlimited.pop(key, None)
        _C._depth_hint[key] = max(2049, int(depth * float(rng.choice([0.3, 0.6, 0.81, 1.0]))))
        _C._SPLIT = "auto"
        with _C.deferred_capacity_check():
            poison()
            o = _C.rasterize_gaussians(sc.bg, sc.means3D, empty, sc.opacities, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                       sc.projmatrix, sc.tanfovx, sc.tanfovy, H, W, sc.shs, 3, sc.campos, False, False)
            poison()
            go = _C.rasterize_gaussians_backward(sc.bg, sc.means3D, o[3], empty, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                                 sc.projmatrix, sc.tanfovx, sc.tanfovy, dc, do, sc.shs, 3, sc.campos, o[4], o[0], o[5], o[6], False)
        ok = _C.check_deferred()
        nlim += 1
        if ok:
            nfit += 1
            if float((o[1] - b[1]).abs().max()) > 3e-5 * max(float(b[1].abs().max()), 1e-20): msgs.append("LIMITED color differs")
            for i, (x, y) in enumerate(zip([t for t in go if t.numel()], b[4])):
                if float((x - y).abs().max()) > 2e-3 * max(float(y.abs().max()), 1e-20): msgs.append(f"LIMITED grad{i} differs")
    if a[0] != b[0]: msgs.append("R")
    for name, x, y in (("color", a[1], b[1]), ("others", a[2], b[2])):
        if not torch.isfinite(y).all(): msgs.append(name + " nonfinite")
        sc_ = max(float(x.abs().max()), 1e-20)
        if float((x - y).abs().max()) > 3e-5 * sc_:
            per = []
            for pl in range(x.shape[0]):
                s1 = max(float(x[pl].abs().max()), 1e-20)
                cnt = int(((x[pl] - y[pl]).abs() > 3e-5 * s1).sum())
                if cnt: per.append((pl, cnt))
            msgs.append(f"{name} {float((x - y).abs().max()) / sc_:.2e} planes(px>tol)={per}")
    mism = float((a[3] != b[3]).float().mean())
    if mism > 1e-4: msgs.append(f"n_contrib {mism:.2e}")
    nflip = int((a[3] != b[3]).sum())
    for i, (x, y) in enumerate(zip(a[4], b[4])):
        if not torch.isfinite(y).all(): msgs.append(f"grad{i} nonfinite")
        sc_ = max(float(x.abs().max()), 1e-20)
        if float((x - y).abs().max()) > 2e-3 * sc_:
            rows = int((((x - y).abs().reshape(x.shape[0], -1).max(dim=1).values) > 2e-3 * sc_).sum())
            msgs.append(f"grad{i} {float((x - y).abs().max()) / sc_:.2e} rows={rows}")
    if msgs:
        bad += 1
        print("MISMATCH", "n_contrib_flips", nflip, dict(N=N, W=W, H=H, radius=radius, kind=str(kind), om=om, sp=sp, maxlen=a[5]), msgs, flush=True)
print(f"fuzz: {n} scenes, {bad} with mismatches, {nlim} segment-limited calls of which {nfit} fit their limit, {time.time() - t0:.0f} s")
