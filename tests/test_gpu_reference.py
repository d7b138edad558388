"""The REAL reference on the GPU: the reference's own rasterizer sources compiled for gfx950
(oracle/ref_build -> oracle/_ref, built where /root/reference exists) run on the MI355X and compared with
(a) the CPU oracle -- this is what pins the oracle -- and (b) the product.

Two builds of the same sources (oracle/ref_build/build_ref.py):
  strict   fp contraction off, rsqrtf = 1/sqrtf: every fp32 operation is the IEEE operation the source names.
           Oracle and product must equal it BIT FOR BIT on every binning integer (radii, tiles touched, sorted
           surfel list, sort keys, tile ranges) at every size, the headline and the 1080p slice included.
  default  hipcc's own FMA contraction and approximate rsqrt (one possible rounding of the reference, as nvcc's
           build is another): ceil(3 * extent) moves by one for a measured fraction of surfels.
Float outputs and `n_contrib` depend on per-(pixel, surfel) threshold tests (alpha >= 1/255, T < 1e-4, T > 0.5,
rho3d <= rho2d) on an ill-conditioned cross product; two roundings of the same formulas flip a MEASURED
fraction of them (the reference's two builds differ from each other by more than either differs from the oracle).
The budgets are FROZEN literals (tests/ref_budgets.py: 2x what round 2 measured); nothing here reads a file a
re-measurement could rewrite."""
import os

import numpy as np
import pytest
import torch

from oracle import surfel_oracle as so
from oracle.ref_build import ref
from tests.ref_budgets import BUDGET, INTEGER, RADIUS_MAX_DELTA
from tests.util import DIST_ATOL, assert_close, look_at_view, make_case, oracle_forward, to_np
from vidu4d_amd.synthetic import make_scene, make_upstream_grads

pytestmark = pytest.mark.gpu
BIG = {  # the configurations of the measured report (tools/ref_parity_report.py CONFIGS)
    "mid": dict(n=5000, width=128, height=128, seed=11),
    "cfgA": dict(n=50_000, width=256, height=256, seed=1234),
    "cfgB": dict(n=200_000, width=512, height=512, seed=1234),          # BASELINE.json headline
    "cfgE_slice": dict(n=250_000, width=1920, height=1080, seed=1234),  # 1080p, partial tiles
    "cfgE_full": dict(n=1_000_000, width=1920, height=1080, seed=1234),  # BASELINE.json configs[4] at its full size
}
GRADS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")


def _need_ref(variant):
    if not ref.available(variant):
        pytest.skip(f"oracle/_ref ({variant}) not built (needs /root/reference at build time)")
    ref.use(variant)


def _budget(config, variant, pair, tensor):
    """(outlier fraction, outlier bound) of this tensor: frozen literals."""
    return BUDGET[config][variant][pair][tensor]


def _check_floats(config, variant, pair, color, others, grads, rf, rg):
    fr, wr = _budget(config, variant, pair, "color")
    assert_close("color", color, rf["color"], outlier_fraction=fr, outlier_rtol=wr)
    for i in range(8):
        fr, wr = _budget(config, variant, pair, f"others{i}")
        if i == 6:   # distortion: a difference of O(1) fp32 sums, compared on its absolute noise floor
            assert_close("others[6]", others[6], rf["others"][6], atol=DIST_ATOL, outlier_fraction=2e-3, outlier_rtol=1.0)
            continue
        # planes 5 / 7 (median depth / weight) are selections: a T > 0.5 flip swaps in another sample
        assert_close(f"others[{i}]", others[i], rf["others"][i], outlier_fraction=fr, outlier_rtol=1.0 if i in (5, 7) else wr)
    for k in GRADS:
        fr, wr = _budget(config, variant, pair, k)
        assert_close(k, grads[k], rg[k], outlier_fraction=fr, outlier_rtol=max(wr, 0.2))


def _n_contrib_mismatch(ref_n, ours):
    """Fraction of differing entries.  A pixel without any contributor keeps the reference's initial
    `float median_contributor = -1` (forward.cu:326), whose conversion to uint32 (:452) is undefined behaviour:
    the default build stores 0, the strict build stores whatever the register holds.  Nobody reads it (the
    backward loop of such a pixel is empty), so the median plane is compared where the pixel has contributors."""
    has = ref_n[0] > 0
    assert np.array_equal(ours[1][~(ours[0] > 0)], np.zeros_like(ours[1][~(ours[0] > 0)])), "ours: 0 without contributors"
    return ((ref_n[0] != ours[0]).sum() + (ref_n[1] != ours[1])[has].sum()) / float(ref_n.size)


def _oracle_vs_ref(config, sc, variant, dev):
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(sc.width, sc.height)
    g = so.backward(st, dc, do)
    d = sc.to(dev)
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dc.to(dev), do.to(dev))
    R, P, W, H = int(rf["num_rendered"]), sc.num_surfels, sc.width, sc.height
    gx, gy = st["grid"]
    radii = to_np(rf["radii"])
    if variant == "strict":
        assert R == st["num_rendered"]
        assert np.array_equal(radii, st["radii"]), "radii"
        assert np.array_equal(ref.state("tiles_touched", P), st["tiles_touched"])
        assert np.array_equal(ref.state("point_list", R), st["point_list"]), "sorted surfel list"
        assert np.array_equal(ref.state("sorted_keys", R), st["point_list_keys"]), "sort keys"
        assert np.array_equal(ref.state("ranges", gx * gy * 2).reshape(-1, 2), st["ranges"]), "tile ranges"
        vis = radii > 0
        assert np.array_equal(ref.state("transMat", P * 9).reshape(P, 9)[vis], st["transMat"][vis]), "homographies"
        assert _n_contrib_mismatch(ref.state("n_contrib", 2 * W * H).reshape(2, H, W), st["n_contrib"]) \
            <= INTEGER[config]["strict"]["oracle_vs_ref"][0], "n_contrib"
    else:
        flips = radii != st["radii"]
        assert flips.mean() <= INTEGER[config]["default"]["oracle_vs_ref"][1] and \
            np.abs(radii.astype(np.int64) - st["radii"]).max() <= 1, f"{int(flips.sum())} radius flips"
    _check_floats(config, variant, "oracle_vs_ref", st["color"], st["others"], g, rf, rg)


@pytest.mark.parametrize("variant", ["strict", "default"])
@pytest.mark.parametrize("case", ["tiny", "ragged", "small", "deg1", "subpixel", "huge", "init_opacity"])
def test_oracle_matches_real_reference(case, variant, gpu_device):
    _need_ref(variant)
    _oracle_vs_ref(case, make_case(case), variant, gpu_device)


@pytest.mark.parametrize("variant", ["strict", "default"])
@pytest.mark.parametrize("config", ["cfgA", "cfgB", "cfgE_slice"])
def test_oracle_matches_real_reference_at_full_sizes(config, variant, gpu_device):
    """50k / 256^2, the 200k / 512^2 headline and a 250k-surfel slice of the 1080p configuration."""
    _need_ref(variant)
    so.set_threads(min(64, os.cpu_count() or 1))
    _oracle_vs_ref(config, make_scene(**BIG[config]), variant, gpu_device)


def _run_product(d, dc, do, W, H):
    from vidu4d_amd import _C
    e = torch.empty(0, device=d.bg.device)
    leaves = [t.clone().requires_grad_(True) for t in (d.means3D, d.opacities, d.scales, d.rotations, d.shs)]
    st = _C.rasterize_gaussians(d.bg, leaves[0], e, leaves[1], leaves[2], leaves[3], 1.0, e, d.viewmatrix, d.projmatrix,
                                d.tanfovx, d.tanfovy, H, W, leaves[4], d.sh_degree, d.campos, False, False)
    R, color, others, radii, geom, binning, img = st
    P, T = d.num_surfels, ((W + 15) // 16) * ((H + 15) // 16)
    rd = lambda what, n: _C.read_state(what, {}, geom, binning, img, P, W, H, torch.int32, n).numpy().view(np.uint32)  # noqa: E731
    ints = dict(radii=radii.cpu().numpy(), R=R, point_list=rd("point_list", R), ranges=rd("ranges", 2 * T).reshape(-1, 2),
                n_contrib=rd("n_contrib", 2 * W * H).reshape(2, H, W))
    import diff_surfel_rasterization as dsr
    rs = dsr.GaussianRasterizationSettings(H, W, d.tanfovx, d.tanfovy, d.bg, 1.0, d.viewmatrix, d.projmatrix,
                                           d.sh_degree, d.campos, False, False)
    m2d = torch.zeros_like(leaves[0], requires_grad=True)
    color, radii2, allmap = dsr.GaussianRasterizer(rs)(means3D=leaves[0], means2D=m2d, opacities=leaves[1],
                                                       shs=leaves[4], scales=leaves[2], rotations=leaves[3])
    torch.autograd.backward([color, allmap], [dc, do])
    grads = dict(zip(("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"), (t.grad for t in leaves)))
    grads["dL_dmeans2D"] = m2d.grad
    return color.detach(), allmap.detach(), grads, ints


@pytest.mark.parametrize("variant", ["strict", "default"])
@pytest.mark.parametrize("config", ["cfgA", "cfgB", "cfgE_slice", "cfgE_full", "object_split", "world_kcam"])
def test_product_matches_real_reference(gpu_device, config, variant, monkeypatch):
    """Product vs reference directly (no oracle in between): at BASELINE's 50k / 256^2 configuration, at the headline
    size, at 1080p (a 250k slice and the FULL largest configuration: 1 M surfels, 8 160 tiles with a partial bottom row,
    13 tile bits -- inside budgets of its own, frozen from round 4's measurement), on an object-centric frame with Stage-3 initialisation opacities blended segment-parallel (lists of
    several thousand entries that never saturate -- the regime the reference walks with one thread block per tile),
    and through a camera that is NOT the Stage-3 one: a rigid world-to-view matrix off the identity and a KCamera
    frustum with an off-centre principal point (gs/scene/cameras.py:106-146: it enters through the fields of view and
    the projection matrix)."""
    _need_ref(variant)
    from vidu4d_amd import _C
    from vidu4d_amd.synthetic import make_object_scene
    if config == "object_split":
        monkeypatch.setattr(_C, "_SPLIT", "1")
        sc = make_object_scene(120_000, 512, radius=0.4, opacity_mode="init")
    elif config == "world_kcam":
        from vidu4d_amd.gs.cameras import KCamera
        cam = KCamera(H=288, W=384, left=-0.42, right=0.55, top=0.31, bottom=-0.40, data_device="cpu")
        tanx, tany = (float(torch.tan(f.float() * 0.5)) for f in (cam.FoVx, cam.FoVy))
        sc = make_scene(60_000, 384, 288, seed=77)
        view, eye = look_at_view((0.4, -0.3, -0.5), (0.0, 0.1, 3.0))
        sc.viewmatrix, sc.campos = view, eye
        sc.projmatrix = (view @ cam.projection_matrix.cpu()).contiguous()
        sc.tanfovx, sc.tanfovy = tanx, tany
    else:
        sc = make_scene(**BIG[config])
    W, H, P = sc.width, sc.height, sc.num_surfels
    d = sc.to(gpu_device)
    dc, do = (t.to(gpu_device) for t in make_upstream_grads(W, H))
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dc, do)
    color, allmap, grads, ints = _run_product(d, dc, do, W, H)
    R = int(rf["num_rendered"])
    if config == "object_split":
        assert R > 200_000  # (long lists: the split is really exercised)
    radii = to_np(rf["radii"])
    # a configuration without a measurement of its own takes the frozen budgets of the nearest measured one:
    # object_split (no saturation, thousands of samples per pixel) those of the 1080p slice
    cfg = {"object_split": "cfgE_slice"}.get(config, config)
    if variant == "strict":
        assert ints["R"] == R and np.array_equal(ints["radii"], radii), "radii"
        assert np.array_equal(ints["point_list"], ref.state("point_list", R)), "sorted surfel list"
        assert np.array_equal(ints["ranges"], ref.state("ranges", ints["ranges"].size).reshape(-1, 2)), "tile ranges"
        budget = 2e-3 if config == "object_split" else INTEGER[cfg]["strict"]["product_vs_ref"][0]
        assert _n_contrib_mismatch(ref.state("n_contrib", 2 * W * H).reshape(2, H, W), ints["n_contrib"]) <= budget, "n_contrib"
    else:
        flips = ints["radii"] != radii
        budget = 1e-2 if config == "object_split" else INTEGER[cfg]["default"]["product_vs_ref"][1]
        assert flips.mean() <= budget and np.abs(ints["radii"].astype(np.int64) - radii).max() <= RADIUS_MAX_DELTA.get(config, 1)
    _check_floats(cfg, variant, "product_vs_ref", color, allmap, grads, rf, rg)


# ---- round 5: gates that are NOT normalised by the tensor's maximum.  relative_error_stats (tests/util.py):
# the relative L2 error of every tensor, with and without the entries a flipped threshold decision moved.  Measured on the
# MI355X (tools/ref_parity_report.py -> profiles/r05_ref_parity.json), rel-L2 WITHOUT the outliers, worst tensor:
#   product vs oracle          50 k / 256^2: 2.0e-5 (dL_dopacity; 1.5e-7 on the planes)    200 k / 512^2: 5.1e-7
#   product vs strict `_ref`   50 k / 256^2: 6.8e-5 (dL_dmeans2D)                           200 k / 512^2: 1.8e-4 (dL_dmeans3D)
#   (oracle vs strict `_ref`: the same numbers to two digits -- it is the reference's rounding, not the product's)
# and WITH them: product vs oracle 1.4e-3 at 50 k (ONE median-sample flip moves dL_dscales of two surfels), 5.1e-7 at 200 k;
# product vs strict 1.4e-3 / 1.2e-3.  The gates below are those figures with a factor of 2.5-5 of head room, frozen.
# Plane 6 (distortion) is compared on its absolute floor above: its rel-L2 is 3e-4 .. 1e-3 between ANY two of the four
# implementations (47 % of its entries differ by more than 1e-4 of scale between the reference's own two builds).
L2_GATE = {"product_vs_oracle": (5e-5, 5e-3), "product_vs_ref": (1e-3, 5e-3)}   # (without outliers, with them)


@pytest.mark.parametrize("config", ["cfgA", "cfgB"])
def test_relative_l2_error_is_gated(gpu_device, config):
    from tests.util import relative_error_stats
    _need_ref("strict")
    so.set_threads(min(64, os.cpu_count() or 1))
    sc = make_scene(**BIG[config])
    W, H = sc.width, sc.height
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(W, H)
    og = so.backward(st, dc, do)
    d = sc.to(gpu_device)
    dcg, dog = dc.to(gpu_device), do.to(gpu_device)
    color, allmap, grads, _ = _run_product(d, dcg, dog, W, H)
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dcg, dog)

    def tensors(c, o, g):
        out = {"color": c}
        out.update({f"others{i}": o[i] for i in range(8) if i != 6})
        out.update({k: g[k] for k in GRADS})
        return out
    ours = tensors(color, allmap, grads)
    for pair, want in (("product_vs_oracle", tensors(st["color"], st["others"], og)),
                       ("product_vs_ref", tensors(rf["color"], rf["others"], rg))):
        trimmed_gate, raw_gate = L2_GATE[pair]
        for name, w in want.items():
            s = relative_error_stats(ours[name], w)
            assert s["rel_l2_without_outliers"] <= trimmed_gate, f"{config} {pair} {name}: rel-L2 without outliers {s}"
            assert s["rel_l2"] <= raw_gate, f"{config} {pair} {name}: rel-L2 {s}"


# ---- round 6: the outliers EXPLAINED, not budgeted --------------------------------------------------
def _replay_check(name, st, dc, do, other_color, other_others, other_n, other_grads, tol=1e-4, strict=0):
    """oracle -> (decisions of the other implementation forced: oracle/decision_replay.py) -> every entry of every tensor
    within `tol` of the tensor's scale.  Returns the explanation's statistics."""
    from oracle import decision_replay as dr
    ex = dr.explain(st, other_color, other_others, other_n, tol=tol, strict=strict)
    assert not ex["unexplained"], (f"{name}: {len(ex['unexplained'])} of {ex['pixels']} disagreeing pixels are not explained by "
                                   f"flipping up to three near-threshold decisions: {ex['unexplained'][:8]}")
    # the pairs blamed sit within rounding of their threshold (relative distance; the candidates are searched up to 2e-3)
    assert ex["max_margin"] <= 2e-4, (name, ex["max_margin"])
    out, g = dr.replay(st, ex, dc, do)
    worst = {}

    def hold(key, got, want, atol=0.0):
        got, want = to_np(got).astype(np.float64), to_np(want).astype(np.float64)
        scale = np.abs(want).max() + 1e-30
        err = np.abs(got - want).max()
        worst[key] = max(0.0, err - atol) / scale
        assert err <= tol * scale + atol, f"{name}: {key} differs by {err / scale:.2e} of scale after the replay ({ex['by_kind']})"
    hold("color", out["color"], other_color)
    for i in range(8):
        hold(f"others{i}", out["others"][i], to_np(other_others)[i], atol=DIST_ATOL if i == 6 else 0.0)
    on = to_np(other_n).astype(np.int64).reshape(2, -1)
    mine = out["n_contrib"].astype(np.int64).reshape(2, -1)
    has = on[0] > 0
    assert np.array_equal(mine[0], on[0]) and np.array_equal(mine[1][has], on[1][has]), f"{name}: n_contrib after the replay"
    for k in GRADS:
        hold(k, g[k], other_grads[k])
    ex["worst_after_replay"] = max(worst.values())
    return ex


@pytest.mark.parametrize("config", ["cfgA", "cfgB", "cfgE_slice"])
def test_every_difference_to_the_strict_reference_is_a_threshold_flip(config, gpu_device):
    """tests/ref_budgets.py COUNTS the entries of the oracle (and the product) that lie beyond 1e-4 of scale from the
    reference's own build; this test EXPLAINS them.  The oracle is re-run with the reference's decisions forced on the
    disagreeing pixels -- its walk end and median sample from its n_contrib, and the accept / branch decision of the few
    (pixel, surfel) pairs within rounding of a threshold inverted (forward.cu:379-405, :416-421; backward.cu:351-353) -- and
    then EVERY entry of EVERY tensor (11 image planes, 6 gradient tensors) agrees with the reference to 1e-4 of scale: the two
    differ by threshold flips and rounding, and by nothing else.  No outlier budget."""
    _need_ref("strict")
    so.set_threads(min(64, os.cpu_count() or 1))
    sc = make_scene(**BIG[config])
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(sc.width, sc.height)
    d = sc.to(gpu_device)
    rf = ref.forward(d)
    rg = ref.backward(d, rf, dc.to(gpu_device), do.to(gpu_device))
    n_ref = ref.state("n_contrib", 2 * sc.width * sc.height).reshape(2, sc.height, sc.width)
    ex = _replay_check(f"oracle vs strict reference, {config}", st, dc, do, to_np(rf["color"]), to_np(rf["others"]), n_ref,
                       {k: to_np(rg[k]) for k in GRADS}, strict=1)
    print(f"decision replay {config}: {ex['pixels']} disagreeing pixels, {ex['by_kind']}, {len(ex['flips'])} flipped pairs, "
          f"largest relative distance to a threshold {ex['max_margin']:.1e}, worst entry after the replay "
          f"{ex['worst_after_replay']:.1e} of scale")


@pytest.mark.parametrize("config", ["cfgA", "cfgB", "cfgE_slice"])
def test_every_difference_of_the_product_to_the_oracle_is_a_threshold_flip(config, gpu_device):
    """The same for the PRODUCT against the oracle: the product's walk end / median sample (its n_contrib) and the few
    near-threshold pair decisions forced into the oracle, then everything within 1e-4 of scale -- planes and gradients."""
    so.set_threads(min(64, os.cpu_count() or 1))
    sc = make_scene(**BIG[config])
    st = oracle_forward(sc)
    dc, do = make_upstream_grads(sc.width, sc.height)
    d = sc.to(gpu_device)
    color, allmap, grads, ints = _run_product(d, dc.to(gpu_device), do.to(gpu_device), sc.width, sc.height)
    ex = _replay_check(f"product vs oracle, {config}", st, dc, do, to_np(color), to_np(allmap), ints["n_contrib"],
                       {k: to_np(grads[k]) for k in GRADS})
    print(f"decision replay (product) {config}: {ex['pixels']} disagreeing pixels, {ex['by_kind']}, worst entry after the replay "
          f"{ex['worst_after_replay']:.1e} of scale")


@pytest.mark.parametrize("radius", [1.0, 0.3])
def test_distortion_plane_against_an_fp64_evaluation(radius, gpu_device):
    """Plane 6 is where the gate against the reference is weakest (an absolute floor): the reference's fp32 form
    m^2 A + M2 - 2 m M1 (forward.cu:411-428) cancels three terms of size ~0.9 to (depth spread)^2 ~ 1e-4; the product keeps the
    moments about a per-tile reference depth.  Against the SAME per-pair alpha / depth carried through in fp64
    (oracle_distortion_f64) the product must be at least as close as the reference's own strict build is -- on object-centric
    scenes (a ball of radius 1.0 / a thin one of 0.3 at distance 3), where the cancellation is worst."""
    from vidu4d_amd.synthetic import make_object_scene
    _need_ref("strict")
    so.set_threads(min(64, os.cpu_count() or 1))
    sc = make_object_scene(60000, 256, 256, radius=radius)
    st = oracle_forward(sc)
    f64 = so.distortion_f64(st)
    d = sc.to(gpu_device)
    dc, do = make_upstream_grads(sc.width, sc.height)
    rf = ref.forward(d)
    _c, allmap, _g, ints = _run_product(d, dc.to(gpu_device), do.to(gpu_device), sc.width, sc.height)
    n_ref = ref.state("n_contrib", 2 * sc.width * sc.height).reshape(2, sc.height, sc.width)
    same = (ints["n_contrib"][0] == st["n_contrib"][0]) & (n_ref[0] == st["n_contrib"][0])   # (same walk: no threshold flip)
    e_prod = np.abs(to_np(allmap[6]).astype(np.float64) - f64)[same]
    e_ref = np.abs(to_np(rf["others"][6]).astype(np.float64) - f64)[same]
    e_orc = np.abs(st["others"][6].astype(np.float64) - f64)[same]
    scale = f64.max()
    print(f"distortion vs fp64, radius {radius}: plane max {scale:.3e}; max |error| product {e_prod.max():.2e}, strict reference "
          f"{e_ref.max():.2e}, oracle (reference's form) {e_orc.max():.2e}; rms {np.sqrt((e_prod ** 2).mean()):.2e} / "
          f"{np.sqrt((e_ref ** 2).mean()):.2e} / {np.sqrt((e_orc ** 2).mean()):.2e}")
    assert same.mean() > 0.99
    assert e_prod.max() <= e_ref.max() and np.sqrt((e_prod ** 2).mean()) <= np.sqrt((e_ref ** 2).mean())
    assert e_prod.max() <= 2e-3 * scale
