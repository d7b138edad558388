"""Paired workgroups for the longest tiles of a whole-tile forward (csrc/blend.hip fwd_pair_walk, VIDU4D_SCHED_PAIR; VERDICT
r5 item 2: two workgroups per tile, each wave on one 8x4 block with its halves on consecutive list entries) against the
one-workgroup walk they stand in for (blend_fwd_kernel = forward.cu:265-463):
  * the transmittance recurrence runs over the two halves' entries in list order with the same roundings: final
    transmittance, contributor counts (the end of the walk, forward.cu:400-405), the median sample (:416-421: contributor,
    depth, weight) and the distortion moments M1 / M2 are BIT-IDENTICAL;
  * colour, depth, normal and distortion sums are even-entries + odd-entries: equal up to fp32 re-association;
  * the recorded segments it leaves serve the same backward: gradients within the float atomics' noise and 1e-5 of scale;
in every blend mode, on frames that saturate early, never, on partial tiles, on long lists, stacked frames included; with
EVERY tile paired (K = 15) and with the rule the product uses (the tiles above K / 4 x the mean list length)."""
import pytest
import torch

from tests.test_gpu_round4 import BLEND_GRADS, _fuzz_scenes, _grad_error, _run
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads

pytestmark = pytest.mark.gpu


def _scene(which):
    if which == "uniform":
        return make_scene(60_000, 256, 256, seed=31)               # lists of ~700 entries: 3 recorded segments per tile
    if which == "saturating":
        sc = make_scene(40_000, 192, 128, seed=32, sigma_px=5.0)
        sc.opacities[:] = 0.9                                      # pixels saturate well inside their lists
        return sc
    if which == "init_opacity":
        return make_scene(40_000, 192, 128, seed=33, opacity_mode="init")   # nothing saturates: walks reach the list ends
    if which == "partial_tiles":
        return make_scene(30_000, 200, 150, seed=34, bg=(0.3, 0.1, 0.6))    # W, H not multiples of 16, coloured background
    if which == "short":
        return make_scene(3_000, 160, 96, seed=35)                 # lists below the recorded segments' minimum
    return make_object_scene(40_000, 256, radius=0.5, opacity_mode="init")  # lists of a few thousand entries


def _planes_close(a, b, what):
    """a: paired workgroups, b: one workgroup per tile"""
    assert torch.equal(a["n_contrib"], b["n_contrib"]), (what, int((a["n_contrib"] != b["n_contrib"]).sum()))
    HW = a["color"].shape[-1] * a["color"].shape[-2]
    fa, fb = a["final_T"].reshape(3, -1), b["final_T"].reshape(3, -1)
    assert fa.shape[1] == HW
    assert torch.equal(fa, fb), (what, "final_T / M1 / M2", [int((fa[i] != fb[i]).sum()) for i in range(3)])
    assert torch.equal(a["radii"], b["radii"])
    # planes: 0 depth, 1 alpha (= 1 - T: exact), 2-4 normal, 5 median depth (exact), 6 distortion, 7 median weight (exact)
    oa, ob = a["others"], b["others"]
    for k in (1, 5, 7):
        assert torch.equal(oa[k], ob[k]), (what, "plane", k, int((oa[k] != ob[k]).sum()))
    for name, x, y in (("color", a["color"], b["color"]), ("depth", oa[0], ob[0]), ("normal", oa[2:5], ob[2:5]),
                       ("distortion", oa[6], ob[6])):
        err = float((x - y).abs().max())
        assert err <= 2e-6 * (float(y.abs().max()) + 1e-30) + 1e-7, (what, name, err, float(y.abs().max()))


@pytest.mark.parametrize("mode", ["full", "lite", "geom"])
@pytest.mark.parametrize("which", ["uniform", "saturating", "init_opacity", "partial_tiles", "short", "object"])
def test_paired_workgroups_equal_the_one_workgroup_walk(gpu_device, monkeypatch, which, mode):
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    monkeypatch.setattr(_C, "_SPLIT", "0")   # (the whole-tile forward whatever earlier frames of this shape suggested)
    sc = _scene(which)
    aux = {"full": 0, "lite": _lib.AUX_ALPHA, "geom": _lib.AUX_GEOM}[mode]
    dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
    if mode != "full":   # the planes the mode does not carry are TAKEN as zero
        keep = [1] if mode == "lite" else [0, 1, 2, 3, 4]
        z = torch.zeros_like(do)
        z[keep] = do[keep]
        do = z
    monkeypatch.setattr(_C, "PAIR_K", 0)
    b = _run(sc, dev, dc, do, aux=aux, flags=0)
    b2 = _run(sc, dev, dc, do, aux=aux, flags=0)
    assert int(b["header"][17]) == 0
    a = _run(sc, dev, dc, do, aux=aux, flags=_lib.sched_pair(15))
    tiles = ((sc.width + 15) // 16) * ((sc.height + 15) // 16)
    assert int(a["header"][17]) == min(tiles, 1024)   # (num_paired: every tile, up to the launch's bound)
    _planes_close(a, b, (which, mode))
    assert int(a["header"][5]) == int(b["header"][5])    # (split_used: both left recorded segments, or neither)
    noise = _grad_error(b2, b, BLEND_GRADS)
    assert _grad_error(a, b, BLEND_GRADS) <= 1e-5 + 4.0 * noise, (which, mode, _grad_error(a, b, BLEND_GRADS), noise)
    assert _grad_error(a, b) <= 1e-4, (which, mode, _grad_error(a, b))
    # ... and from the one-workgroup-per-tile backward (no recorded segments read)
    c = _run(sc, dev, dc, do, aux=aux, flags=_lib.sched_pair(15) | _lib.DEBUG_WHOLE_TILE_BACKWARD)
    d = _run(sc, dev, dc, do, aux=aux, flags=_lib.DEBUG_WHOLE_TILE_BACKWARD)
    assert _grad_error(c, d, BLEND_GRADS) <= 1e-5 + 4.0 * noise, (which, mode, _grad_error(c, d, BLEND_GRADS), noise)


@pytest.mark.parametrize("seed,large,n", [(3, False, 16), (11, True, 6)])
def test_paired_workgroups_on_the_fuzz_scenes_with_and_without_culls(gpu_device, monkeypatch, seed, large, n):
    """The scenes that broke round 3's cull (huge, foreshortened, near-plane surfels), image sizes up to 1920 x 1080 (more
    tiles than the launch pairs: both walks in one launch): the paired walk with the culls off evaluates every entry for
    every block -- the reference's walk -- and must stop where the culled walk stops."""
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    monkeypatch.setattr(_C, "PAIR_K", 0)
    for sc, what in _fuzz_scenes(n, seed, large):
        dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
        b = _run(sc, dev, dc, do, flags=0)
        a = _run(sc, dev, dc, do, flags=_lib.sched_pair(15))
        n8 = _run(sc, dev, dc, do, flags=_lib.sched_pair(15) | _lib.DEBUG_NO_CULL)
        _planes_close(a, b, what)
        _planes_close(n8, b, what)   # (which half takes an entry depends on the culls: the sums' association, nothing else)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_the_pairing_rule_takes_the_tiles_above_a_multiple_of_the_mean(gpu_device, monkeypatch, k):
    """K / 4 x the mean list length, by length class (1024 classes of the longest list): num_paired lies between the counts
    of the tiles above the thresholds one class to either side; the paired tiles are the front of the schedule; two stacked
    frames through the public op run the same rule over both frames' tiles."""
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    monkeypatch.setattr(_C, "_SPLIT", "0")
    monkeypatch.setattr(_C, "PAIR_K", 0)
    sc = make_object_scene(40_000, 256, radius=0.5, opacity_mode="init")
    dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
    b = _run(sc, dev, dc, do, flags=0)
    a = _run(sc, dev, dc, do, flags=_lib.sched_pair(k))
    _planes_close(a, b, k)
    d = sc.to(dev)
    W, H, P = d.width, d.height, d.num_surfels
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    # (the lists' lengths from the contributor counts' upper bound are not available; from the ranges)
    e = torch.empty(0, device=dev)
    with _C.debug_flags(_lib.sched_pair(k)):
        out = _C.rasterize_gaussians(d.bg, d.means3D, e, d.opacities, d.scales, d.rotations, 1.0, e, d.viewmatrix, d.projmatrix,
                                     d.tanfovx, d.tanfovy, d.height, d.width, d.shs, d.sh_degree, d.campos, False, False)
    R, _c, _o, _r, geom, binning, img = out
    ranges = _C.read_state("ranges", None, geom, binning, img, P, W, H, torch.int32, 2 * tiles).cpu().numpy().reshape(tiles, 2)
    order = _C.read_state("tile_order", None, geom, binning, img, P, W, H, torch.int32, tiles).cpu().numpy()
    lens = (ranges[:, 1] - ranges[:, 0]).astype("int64")
    hdr = geom[:256].view(torch.int32).cpu().numpy()
    n = int(hdr[17])
    thr = int(lens.sum()) * k // (4 * tiles)
    cls_width = (int(lens.max()) + 1) / 1024.0
    assert 0 < n < tiles
    assert (lens > thr + 2 * cls_width).sum() <= n <= (lens > thr - 2 * cls_width).sum(), (n, thr, cls_width)
    assert lens[order[:n]].min() >= lens[order[n:]].max() - cls_width   # the front of the schedule, up to a class


def test_paired_workgroups_stacked_frames_equal_per_frame_calls(gpu_device, monkeypatch):
    """Two stacked frames through the public op (the Stage-3 step's form), every tile paired: the same frames rendered one
    by one in the same way (a tile's walk does not depend on what else the launch holds)."""
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C, _lib
    from vidu4d_amd.synthetic import frame_motion
    dev = gpu_device
    monkeypatch.setattr(_C, "_SPLIT", "0")
    sc = make_object_scene(30_000, 192, radius=0.7, opacity_mode="init").to(dev)
    frames = [frame_motion(sc, f, 16) for f in (2, 9)]
    rs = dsr.GaussianRasterizationSettings(sc.height, sc.width, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                           sc.sh_degree, sc.campos, False, False)
    m = torch.stack([f.means3D for f in frames])
    r = torch.stack([f.rotations for f in frames])
    with _C.debug_flags(_lib.sched_pair(15)), torch.no_grad():
        color, radii, others = dsr.rasterize_frames(m, torch.zeros_like(m), sc.shs, sc.opacities, sc.scales, r, [rs, rs])[:3]
        for i, f in enumerate(frames):
            rast = dsr.GaussianRasterizer(rs)
            c1, _r1, o1 = rast(means3D=f.means3D, means2D=torch.zeros_like(f.means3D), shs=sc.shs, opacities=sc.opacities,
                               scales=sc.scales, rotations=f.rotations)[:3]
            # (which tiles leave recorded segments follows the launch's longest list: the sums' association may differ
            # between the stacked and the single launch -- with or without pairs --, what the walk decides may not)
            for k in (1, 5, 7):
                assert torch.equal(others[k, i], o1[k]), (i, k)
            assert float((color[:, i] - c1).abs().max()) <= 1e-6 and float((others[:, i] - o1).abs().max()) <= 1e-6 * (1 + float(o1.abs().max()))


@pytest.mark.parametrize("which,k", [("concentrated", 6), ("concentrated_opaque", 15), ("mixed", 15), ("partial_tiles", 6)])
def test_paired_workgroups_match_the_oracle(gpu_device, monkeypatch, which, k):
    """The paired walk against the CPU ORACLE (forward.cu:265-463 / backward.cu:143-449 restated, oracle/surfel_oracle.c), not only
    against the product's own one-workgroup walk: the scenes of the segment-parallel path's oracle tests (long lists, pixels
    that saturate inside them, tiles at the image border), whole tiles, with the product's pairing rule (K = 6) and with
    every tile paired.  Same checks as every forward / backward parity test: integers identical, floats within the usual
    tolerance, the recorded backward fed from the paired forward included."""
    import numpy as np
    from tests.test_gpu_parity import _check_backward, _check_forward, _concentrated, _native_forward
    from tests.util import oracle_forward
    from vidu4d_amd import _C, _lib
    monkeypatch.setattr(_C, "_SPLIT", "0")
    monkeypatch.setattr(_C, "PAIR_K", k)
    if which == "concentrated":
        sc = _concentrated()
    elif which == "concentrated_opaque":
        sc = _concentrated(seed=92)
        sc.opacities[:] = 0.6
    elif which == "mixed":
        sc = make_scene(60_000, 256, 192, seed=95, sigma_px=6.0)
    else:
        sc = make_scene(20_000, 150, 121, seed=96, sigma_px=1.0)
        g = torch.Generator().manual_seed(96)
        sc.means3D[:, 0] = (torch.rand(20_000, generator=g) - 0.2) * 0.5 * sc.means3D[:, 2]
        sc.means3D[:, 1] = (torch.rand(20_000, generator=g) - 0.2) * 0.5 * sc.means3D[:, 2]
        sc.opacities[:] = 0.15
    st = oracle_forward(sc)
    d, shs, cols, out = _native_forward(sc, gpu_device)
    geom = out[4]
    n_paired = int(geom[:256].view(torch.int32)[17])
    assert n_paired > 0, "the scene does not exercise the pairs"
    _check_forward(sc, st, out)
    _check_backward(sc, st, d, shs, cols, out, gpu_device)
