"""Fuzz (GPU box): the alpha-only split blend WITHOUT its transmittance pre-pass (relative segments + the combine's repair
of the saturating segment, the default) against the one WITH it (VIDU4D_SURFEL_SPEC=0 semantics) on random scenes --
sizes, resolutions, object radii, opacities, footprints -- with recycled allocator blocks poisoned by NaNs.
Contributor counts must agree except for a handful of threshold flips, colour / alpha to 3e-6 of scale, gradients to
5e-4 of scale.  Usage: python tools/fuzz_relative_segments.py [seed] [seconds]"""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
empty = torch.empty(0, device=dev)


def poison():
    blocks = [torch.full((16 * 1024 * 1024,), float("nan"), device=dev) for _ in range(4)]
    del blocks


def run(sc, dc, d_alpha, spec):
    _C._SPLIT, _C._SPEC = "1", spec
    poison()
    out = _C.rasterize_gaussians(sc.bg, sc.means3D, empty, sc.opacities, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                 sc.projmatrix, sc.tanfovx, sc.tanfovy, sc.height, sc.width, sc.shs, 3, sc.campos, False, False,
                                 aux_planes=_lib.AUX_ALPHA)
    R, color, others, radii, geom, binning, img = out
    do = torch.zeros(8, sc.height, sc.width, device=dev)
    do[1] = d_alpha
    poison()
    g = _C.rasterize_gaussians_backward(sc.bg, sc.means3D, radii, empty, sc.scales, sc.rotations, 1.0, empty, sc.viewmatrix,
                                        sc.projmatrix, sc.tanfovx, sc.tanfovy, dc, do, sc.shs, 3, sc.campos, geom, R, binning,
                                        img, False, aux_planes=_lib.AUX_ALPHA)
    hdr = geom[:64].view(torch.int32).cpu()
    ncon = _C.read_state("n_contrib", None, geom, binning, img, sc.num_surfels, sc.width, sc.height, torch.int32,
                         2 * sc.width * sc.height)[: sc.width * sc.height]
    return color, others[1], ncon, [t for t in g if t.numel()], hdr


t0, n, bad, n_split, n_sat = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    N = int(rng.choice([6000, 20000, 60000, 150000]))
    W = int(rng.choice([64, 96, 130, 256, 400])); H = int(rng.choice([48, 80, 128, 256, 333]))
    kind = rng.choice(["object", "uniform", "packed"])
    sp = float(rng.choice([0.7, 1.5, 4.0]))
    seed = int(rng.integers(1 << 30))
    if kind == "object":
        sc = make_object_scene(N, W, H, radius=float(rng.choice([0.1, 0.25, 0.5])), seed=seed, sigma_px=sp)
    else:
        sc = make_scene(N, W, H, seed=seed, sigma_px=sp)
        if kind == "packed":
            sc.means3D[:, :2] *= float(rng.choice([0.1, 0.2, 0.4]))   # everything into a few tiles: long lists
    if rng.random() < 0.8:
        sc.opacities[:] = float(rng.choice([0.003, 0.01, 0.03, 0.1, 0.3, 0.9]))
    sc = sc.to(dev)
    dc, do = make_upstream_grads(W, H, seed=int(rng.integers(1 << 30)))
    dc, d_alpha = dc.to(dev), do[1].to(dev)
    a = run(sc, dc, d_alpha, False)
    b = run(sc, dc, d_alpha, True)
    n += 1
    n_split += int(a[4][3]) > 0
    n_sat += a[4][8:9].view(torch.float32).item() < 1.001e-4 and int(a[4][3]) > 0
    msgs = []
    if int(a[4][6]) or int(b[4][6]):
        msgs.append(f"truncated {int(a[4][6])} {int(b[4][6])}")
    flips = int((a[2] != b[2]).sum())
    if flips > max(3, int(2e-5 * W * H)):
        msgs.append(f"n_contrib differs at {flips} pixels")
    for name, x, y in (("colour", a[0], b[0]), ("alpha", a[1], b[1])):
        if not torch.isfinite(y).all():
            msgs.append(name + " not finite")
        # (pixels whose last contributor flipped carry the flip's weight, < 1e-4)
        tol = 3e-6 * max(float(x.abs().max()), 1e-20) + (2e-4 if flips else 0.0)
        if float((x - y).abs().max()) > tol:
            msgs.append(f"{name} {float((x - y).abs().max()):.2e} > {tol:.1e}")
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        if not torch.isfinite(y).all():
            msgs.append(f"grad{i} not finite")
        elif float((x - y).abs().max()) > 5e-4 * max(float(x.abs().max()), 1e-20):
            msgs.append(f"grad{i} {float((x - y).abs().max()) / max(float(x.abs().max()), 1e-20):.2e}")
    if msgs:
        bad += 1
        print(f"MISMATCH N={N} {W}x{H} {kind} sigma={sp} seed={seed} opac={float(sc.opacities[0]):.3f}:", "; ".join(msgs), flush=True)
print(f"{n} scenes, {n_split} with split tiles, {n_sat} of those saturating, {bad} mismatching")
sys.exit(1 if bad else 0)
