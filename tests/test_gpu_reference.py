"""The REAL reference on the GPU: the reference's own rasterizer sources compiled for gfx950
(oracle/ref_build -> oracle/_ref/libref_surfel.so, built where /root/reference exists) run on the
MI355X and compared with (a) the CPU oracle -- this is what pins the oracle -- and (b) the product.
The committed fixtures the reference produced (tests/golden/ref_*.npz) are checked on the CPU in
tests/test_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import surfel_oracle as so
from oracle.ref_build import ref
from tests.util import DIST_ATOL, assert_close, make_case, oracle_forward, to_np
from vidu4d_amd.synthetic import make_upstream_grads

pytestmark = pytest.mark.gpu

# Reference-vs-anything comparisons differ by how hipcc contracts the reference's arithmetic into
# FMAs (unspecified, like nvcc's), by its approximate rsqrt and by atomic order: threshold flips are
# ~100x more frequent than between oracle and product, hence the wider outlier budget.
REF_OUTLIERS = 2e-3


def _need_ref():
    if not ref.available():
        pytest.skip("oracle/_ref/libref_surfel.so not built (needs /root/reference at build time)")


def _compare(tag, st, g, rf, rg, sc):
    """Integer outputs must be identical unless a surfel's radius sits on a ceil() boundary: the
    reference as compiled by hipcc contracts its fp32 arithmetic into FMAs and uses an approximate
    rsqrt, which moves ceil(3*extent) by one for a handful of surfels (nvcc would do the same in its
    own way).  Those cases are bounded (<= 0.1 % of the surfels, |delta radius| <= 1) and then only
    the images and gradients are compared, with a wider outlier budget."""
    R = rf["num_rendered"]
    P, W, H = sc.num_surfels, sc.width, sc.height
    gx, gy = st["grid"]
    radii = to_np(rf["radii"])
    flips = radii != st["radii"]
    integers_exact = not flips.any()
    budget = REF_OUTLIERS
    if not integers_exact:
        assert flips.mean() <= 1e-3 and np.abs(radii - st["radii"]).max() <= 1, f"{tag}: {flips.sum()} radius flips"
        budget = 2e-2
    if integers_exact:
        assert R == st["num_rendered"], f"{tag}: num_rendered {R} != {st['num_rendered']}"
        assert np.array_equal(radii, st["radii"]), f"{tag}: radii"
        assert np.array_equal(ref.state("tiles_touched", P), st["tiles_touched"])
        assert np.array_equal(ref.state("point_list", R), st["point_list"]), f"{tag}: sorted surfel list"
        assert np.array_equal(ref.state("sorted_keys", R), st["point_list_keys"])
        assert np.array_equal(ref.state("ranges", gx * gy * 2).reshape(-1, 2), st["ranges"])
        vis = radii > 0
        assert_close(f"{tag}:transMat", ref.state("transMat", P * 9).reshape(P, 9)[vis], st["transMat"][vis], rtol=1e-5,
                     outlier_fraction=REF_OUTLIERS)
        assert_close(f"{tag}:rgb", ref.state("rgb", P * 3).reshape(P, 3)[vis], st["rgb"][vis], rtol=1e-5,
                     outlier_fraction=REF_OUTLIERS)
        ncon = ref.state("n_contrib", 2 * W * H).reshape(2, H, W)
        assert (ncon != st["n_contrib"]).mean() <= REF_OUTLIERS, f"{tag}: n_contrib"
    assert_close(f"{tag}:color", rf["color"], st["color"], outlier_fraction=budget)
    for i in range(8):
        # planes 5 / 7 (median depth / weight) are selections: a T > 0.5 flip swaps in another sample
        assert_close(f"{tag}:others[{i}]", rf["others"][i], st["others"][i], atol=DIST_ATOL if i == 6 else 0.0,
                     outlier_fraction=budget, outlier_rtol=1.0 if i in (5, 7) else 5e-2)
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dtransMat",
              "dL_dcolors"):
        assert_close(f"{tag}:{k}", rg[k], g[k], outlier_fraction=budget)
    return integers_exact


@pytest.mark.parametrize("case", ["tiny", "ragged", "small", "deg1", "subpixel", "huge", "init_opacity"])
def test_oracle_matches_real_reference(case, gpu_device):
    _need_ref()
    sc = make_case(case)
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(sc.width, sc.height)
    g = so.backward(st, dc, do)
    d = sc.to(gpu_device)
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dc.to(gpu_device), do.to(gpu_device))
    _compare(case, st, g, rf, rg, sc)


@pytest.mark.parametrize("scene", ["headline", "object_split"])
def test_product_matches_real_reference_headline(gpu_device, scene, monkeypatch):
    """Product vs reference directly (no oracle in between): at the headline size, and on an
    object-centric frame with Stage-3 initialisation opacities blended segment-parallel (lists of several
    thousand entries that never saturate -- the regime the reference walks with one thread block per tile)."""
    _need_ref()
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C
    from vidu4d_amd.synthetic import make_object_scene, make_scene
    if scene == "headline":
        sc = make_scene(200_000, 512)
    else:
        monkeypatch.setattr(_C, "_SPLIT", "1")
        sc = make_object_scene(120_000, 512, radius=0.4, opacity_mode="init")
    d = sc.to(gpu_device)
    dc, do = (t.to(gpu_device) for t in make_upstream_grads(512, 512))
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dc, do)
    rs = dsr.GaussianRasterizationSettings(512, 512, sc.tanfovx, sc.tanfovy, d.bg, 1.0, d.viewmatrix, d.projmatrix, 3,
                                           d.campos, False, False)
    leaves = [t.clone().requires_grad_(True) for t in (d.means3D, d.opacities, d.scales, d.rotations, d.shs)]
    m2d = torch.zeros_like(leaves[0], requires_grad=True)
    color, radii, allmap = dsr.GaussianRasterizer(rs)(means3D=leaves[0], means2D=m2d, opacities=leaves[1],
                                                      shs=leaves[4], scales=leaves[2], rotations=leaves[3])
    torch.autograd.backward([color, allmap], [dc, do])
    if scene == "object_split":
        assert int(rf["num_rendered"]) > 200_000  # (long lists: the split is really exercised)
    # At 512x512 the reference's AABB extent (h = sqrt(c^2 - ...), forward.cu:152-159) cancels ~4 digits
    # (c ~ 256 px, h ~ 3 px), so the FMA contraction hipcc applies to the reference moves ceil(3h) by
    # one for a few surfels per thousand; the product follows the oracle's fixed operation order.
    flips = radii != rf["radii"]
    assert float(flips.float().mean()) <= 1e-2 and int((radii - rf["radii"]).abs().max()) <= 1, \
        f"{int(flips.sum())} radius flips vs the reference"
    budget = 2e-2 if bool(flips.any()) else REF_OUTLIERS
    assert_close("color", color, rf["color"], outlier_fraction=budget)
    for i in range(8):
        assert_close(f"others[{i}]", allmap[i], rf["others"][i], atol=DIST_ATOL if i == 6 else 0.0,
                     outlier_fraction=budget, outlier_rtol=1.0 if i in (5, 7) else 5e-2)
    for t, k in zip(leaves + [m2d], ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh",
                                      "dL_dmeans2D")):
        assert_close(k, t.grad, rg[k], outlier_fraction=budget)
