"""Shared helpers of the parity tests."""
from __future__ import annotations

import numpy as np
import torch

from oracle import surfel_oracle as so
from vidu4d_amd.synthetic import SurfelScene, make_scene, make_upstream_grads

# Tolerances (BASELINE.json north_star: bit-exact on tile-bin / sort indices, 1e-4 relative on
# rendered RGB / depth / normal and their gradients).  "Relative" is taken against the largest
# magnitude of the compared array (per output plane / per gradient tensor), the usual meaning for
# accumulated fp32 quantities whose individual entries pass through zero.
RTOL = 1e-4
# Product vs the CPU oracle (the same operation sequence wherever a threshold depends on it; rcp / exp and the summation
# order differ): 1e-5 of the tensor's scale -- the measured worst is ~1.5e-6 (profiles/r04_ref_parity.json,
# product_vs_oracle), so a regression of one order of magnitude trips it.  RTOL above stays the bar against the reference's
# own builds (`_ref`), whose two roundings differ from each other by more.
ORACLE_RTOL = 1e-5
# A (pixel, surfel) pair whose alpha or transmittance sits within one ulp of a threshold
# (alpha < 1/255, T < 1e-4, rho3d <= rho2d, T > 0.5) may fall on different sides in two fp32
# implementations that differ only in rounding (exp, FMA contraction, rcp).  Such a flip changes one
# pixel by up to ~0.4 % of a colour value; they are rare (~1e-7 per pair) and unavoidable between
# any two implementations, the reference's own CUDA build included.  The budget below bounds them.
OUTLIER_FRACTION = 2e-5
OUTLIER_RTOL = 5e-2
# the distortion plane is a difference of O(1) fp32 sums: absolute fp32 noise floor
DIST_ATOL = 2e-6


def to_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def assert_close(name, got, want, rtol=RTOL, atol=0.0, outlier_fraction=OUTLIER_FRACTION, outlier_rtol=OUTLIER_RTOL,
                 min_outliers=None):
    got = to_np(got).astype(np.float64)
    want = to_np(want).astype(np.float64)
    assert got.shape == want.shape, f"{name}: shape {got.shape} != {want.shape}"
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    scale = np.abs(want).max() if want.size else 0.0
    err = np.abs(got - want)
    tol = rtol * scale + atol
    bad = err > tol
    frac = bad.mean() if bad.size else 0.0
    worst = err.max() / (scale + 1e-30) if err.size else 0.0
    # one flipped (pixel, surfel) pair touches one pixel of an image plane, or one surfel's row of a
    # per-surfel tensor: that many entries are always tolerated when the fraction budget is non-zero
    if min_outliers is None:
        if got.ndim == 3 and got.shape[0] <= 8:      # (C,H,W) image: one pixel
            min_outliers = got.shape[0]
        elif got.ndim == 2 and got.shape[1] > 64:    # (H,W) plane
            min_outliers = 1
        elif got.ndim >= 2:                          # per-surfel tensor: one row
            min_outliers = int(np.prod(got.shape[1:]))
        else:
            min_outliers = 1
    allowed = max(int(np.ceil(outlier_fraction * bad.size)), min_outliers) if outlier_fraction > 0 else 0
    assert bad.sum() <= allowed, (f"{name}: {bad.sum()} of {bad.size} entries ({frac:.2e}) exceed rtol {rtol:g} "
                                  f"(scale {scale:.3e}, worst {worst:.3e}, allowed {allowed})")
    assert (err <= outlier_rtol * scale + atol).all(), f"{name}: worst error {worst:.3e} beyond the outlier bound"
    return worst


def relative_error_stats(got, want, floor=1e-3):
    """Beyond "a fraction of the tensor's max": (i) the relative L2 error ||got - want|| / ||want|| and (ii) the element-wise
    relative error |got - want| / |want| over the entries whose reference magnitude is above `floor` of the tensor's
    scale (median, 99th percentile, maximum) -- a uniformly small-magnitude region of a tensor cannot drift unseen behind
    a max-normalised gate."""
    got, want = to_np(got).astype(np.float64), to_np(want).astype(np.float64)
    scale = np.abs(want).max() if want.size else 0.0
    nrm = float(np.sqrt((want * want).sum()))
    err = got - want
    out = {"rel_l2": float(np.sqrt((err ** 2).sum()) / (nrm + 1e-300)) if want.size else 0.0}
    # ... and without the entries a flipped threshold decision moved (those beyond 1e-4 of the scale: the set the outlier
    # budgets of assert_close bound by count and size): what is left is rounding, and must be small in the L2 sense too
    keep = np.abs(err) <= RTOL * scale
    out["rel_l2_without_outliers"] = float(np.sqrt((err[keep] ** 2).sum()) / (nrm + 1e-300)) if want.size else 0.0
    big = np.abs(want) > floor * scale
    if big.any():
        r = np.abs(got - want)[big] / np.abs(want)[big]
        out.update(elem_rel_p50=float(np.percentile(r, 50)), elem_rel_p99=float(np.percentile(r, 99)), elem_rel_max=float(r.max()),
                   elem_rel_entries=int(big.sum()), elem_rel_floor=floor)
    return out


def oracle_forward(sc: SurfelScene, colors_precomp=None, stats=False):
    kw = dict(shs=sc.shs) if colors_precomp is None else dict(colors_precomp=colors_precomp)
    return so.forward(sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.viewmatrix, sc.projmatrix, sc.campos,
                      sc.bg, sc.width, sc.height, sc.tanfovx, sc.tanfovy, sc.sh_degree, stats=stats, **kw)


def look_at_view(eye, target, up=(0.0, 1.0, 0.0)):
    """A non-trivial world->view matrix in the reference's row-vector (transposed) convention."""
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    up = torch.tensor(up, dtype=torch.float64)
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(up, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z], 0)  # rows: camera axes in world coordinates
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3] = R
    V[:3, 3] = -R @ eye
    return V.t().contiguous().float(), eye.float()


CASES = {
    # name: (kwargs for make_scene, description)
    "tiny": dict(n=64, width=32, height=32, seed=5),
    "ragged": dict(n=3000, width=70, height=50, seed=7, bg=(0.2, 0.5, 0.7)),  # W,H not multiples of 16
    "small": dict(n=5000, width=128, height=128, seed=11),
    "deg0": dict(n=2000, width=64, height=64, seed=13, sh_degree=0),
    "deg1": dict(n=2000, width=64, height=64, seed=14, sh_degree=1),
    "deg2": dict(n=2000, width=64, height=64, seed=15, sh_degree=2),
    "init_opacity": dict(n=4000, width=96, height=96, seed=17, opacity_mode="init"),
    "subpixel": dict(n=4000, width=96, height=96, seed=19, sigma_px=0.15),  # low-pass filter branch dominates
    "huge": dict(n=300, width=96, height=80, seed=23, sigma_px=20.0, big_fraction=0.1),  # screen-filling surfels
}


def make_case(name, device="cpu"):
    return make_scene(device=device, **CASES[name])
