"""Round-4 GPU parity tests:
  * the footprint cull A/B on the device (VIDU4D_DEBUG_NO_CULL: every list entry evaluated for every pixel, the reference's
    walk, forward.cu:359-405): forward planes, final_T and n_contrib BIT-IDENTICAL with and without the culls, gradients
    within the noise of their float atomics -- on the CPU fuzz generator's scenes, the seeds that broke round 3 included;
  * recorded segments (the backward of a whole-tile forward runs segment-parallel from the per-pixel sums the forward's
    walk stores every 256 entries) against the one-workgroup-per-tile backward, in every blend mode;
  * the GEOM blend (aux_planes = planes 0-4) against the full blend with zero upstream on planes 5-7."""
import numpy as np
import pytest
import torch

from tests.util import make_case
from vidu4d_amd.synthetic import make_object_scene, make_scene, make_upstream_grads

pytestmark = pytest.mark.gpu
GRAD_NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dtransMat", "dL_dsh", "dL_dscales",
              "dL_drotations")


def _run(sc, dev, dc, do, aux=0, flags=None, frames=None):
    """One forward + backward through the native functions; returns images, integer state, gradients, header."""
    from vidu4d_amd import _C
    d = sc.to(dev) if sc.means3D.device.type == "cpu" else sc
    e = torch.empty(0, device=dev)
    ctx = _C.debug_flags(flags) if flags is not None else _C.debug_flags(_C.DEBUG_FLAGS)
    with ctx:
        out = _C.rasterize_gaussians(d.bg, d.means3D, e, d.opacities, d.scales, d.rotations, 1.0, e, d.viewmatrix,
                                     d.projmatrix, d.tanfovx, d.tanfovy, d.height, d.width, d.shs, d.sh_degree, d.campos,
                                     False, False, aux_planes=aux)
        R, color, others, radii, geom, binning, img = out
        g = _C.rasterize_gaussians_backward(d.bg, d.means3D, radii, e, d.scales, d.rotations, 1.0, e, d.viewmatrix,
                                            d.projmatrix, d.tanfovx, d.tanfovy, dc, do, d.shs, d.sh_degree, d.campos, geom,
                                            R, binning, img, False, aux_planes=aux)
    W, H, P = d.width, d.height, d.num_surfels
    ncon = _C.read_state("n_contrib", None, geom, binning, img, P, W, H, torch.int32, 2 * W * H)
    fT = _C.read_state("final_T", None, geom, binning, img, P, W, H, torch.float32, 3 * W * H)
    header = geom[:256].view(torch.int32).cpu()   # (the 64 words of surfel_state.h Header)
    return dict(color=color, others=others, radii=radii, n_contrib=ncon, final_T=fT, grads=dict(zip(GRAD_NAMES, g)),
                header=header, R=R)


# what the blend kernel itself accumulates (float atomics); the other tensors are preprocess_bwd's deterministic chain rule
# applied to these, which can amplify a rounding difference when its terms cancel (dL_dscales of a near-degenerate
# surfel: tools/fuzz_footprint_gpu.py prices that separately)
BLEND_GRADS = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dtransMat")


def _grad_error(a, b, names=GRAD_NAMES):
    """largest |a - b| / max|b| over the gradient tensors"""
    worst = 0.0
    for k in names:
        x, y = a["grads"][k], b["grads"][k]
        assert torch.isfinite(x).all(), k
        worst = max(worst, float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30))
    return worst


def _grads_close(a, b, tol, what=""):
    err = _grad_error(a, b)
    assert err <= tol, (what, err)


def _fuzz_scenes(n, seed, large=False):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_scenes.py")
    spec = importlib.util.spec_from_file_location("fuzz_scenes", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(seed)
    return [mod.random_scene(rng, large) for _ in range(n)]


@pytest.mark.parametrize("seed,large,n", [(3, False, 24), (0, False, 12), (11, True, 8)])
def test_footprint_cull_ab_on_the_device(gpu_device, monkeypatch, seed, large, n):
    """Forward planes, final_T, n_contrib bit-identical with and without the culls.  The gradient sums are float atomics
    whose order differs from run to run (screen-filling surfels add thousands of terms of both signs: the same kernel run
    twice differs by up to ~5e-6 of the tensor's scale on these scenes), so they are held to 1e-6 of scale plus four times
    the noise floor measured on the spot (culls on, twice) -- on the sums the blend kernel makes (BLEND_GRADS).  seed 3: the scenes whose footprints round 3's first conic test
    cut (huge, strongly foreshortened, near-plane surfels); `large`: image sizes up to 1920 x 1080."""
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    # (one workgroup per tile: in a paired tile the culls decide which half of a wave takes an entry, i.e. the association of
    # its sums -- tests/test_gpu_paired_tiles.py holds the paired walk to "stops where the culled walk stops")
    monkeypatch.setattr(_C, "PAIR_K", 0)
    for sc, what in _fuzz_scenes(n, seed, large):
        dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
        a = _run(sc, dev, dc, do, flags=0)
        a2 = _run(sc, dev, dc, do, flags=0)
        b = _run(sc, dev, dc, do, flags=_lib.DEBUG_NO_CULL)
        for k in ("color", "others", "radii", "n_contrib", "final_T"):
            assert torch.equal(a[k], b[k]), (what, k, int((a[k] != b[k]).sum()))
        noise = _grad_error(a2, a, BLEND_GRADS)
        assert _grad_error(a, b, BLEND_GRADS) <= 1e-6 + 4.0 * noise, (what, _grad_error(a, b, BLEND_GRADS), noise)
        assert _grad_error(a, b) <= 1e-4, (what, _grad_error(a, b))   # the chain-rule tensors: the north star's tolerance


@pytest.mark.parametrize("mode", ["full", "lite", "geom"])
@pytest.mark.parametrize("which", ["uniform", "saturating", "init_opacity", "partial_tiles", "object"])
def test_recorded_segments_equal_the_whole_tile_backward(gpu_device, monkeypatch, which, mode):
    """Same forward (recording costs it nothing it computes), backward from the recorded segments against the backward
    that walks every tile with one workgroup (VIDU4D_DEBUG_WHOLE_TILE_BACKWARD): gradients equal up to fp32
    re-association of the forward's running sums (final - prefix instead of the back-to-front recurrence)."""
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    monkeypatch.setattr(_C, "_SPLIT", "0")   # (whole-tile forward whatever earlier frames of this shape suggested)
    if which == "uniform":
        sc = make_scene(60_000, 256, 256, seed=31)               # lists of ~700 entries: 3 segments per tile
    elif which == "saturating":
        sc = make_scene(40_000, 192, 128, seed=32, sigma_px=5.0)
        sc.opacities[:] = 0.9                                    # pixels saturate well inside their lists
    elif which == "init_opacity":
        sc = make_scene(40_000, 192, 128, seed=33, opacity_mode="init")   # nothing saturates: walks reach the list ends
    elif which == "partial_tiles":
        sc = make_scene(30_000, 200, 150, seed=34, bg=(0.3, 0.1, 0.6))    # W, H not multiples of 16, coloured background
    else:
        sc = make_object_scene(40_000, 256, radius=0.5, opacity_mode="init")  # lists of a few thousand entries
    aux = {"full": 0, "lite": _lib.AUX_ALPHA, "geom": _lib.AUX_GEOM}[mode]
    dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
    if mode != "full":   # the planes the mode does not carry are TAKEN as zero
        keep = [1] if mode == "lite" else [0, 1, 2, 3, 4]
        z = torch.zeros_like(do)
        z[keep] = do[keep]
        do = z
    rec = _run(sc, dev, dc, do, aux=aux, flags=0)
    whole = _run(sc, dev, dc, do, aux=aux, flags=_lib.DEBUG_WHOLE_TILE_BACKWARD)
    assert int(rec["header"][5]) == 2 and int(rec["header"][3]) > 0, "no recorded segments"   # split_used, num_segments
    # the backward's workgroups: one per full segment the forward's walk REACHED (Header::num_live_full, word 13) -- on the
    # saturating scene far fewer than the full segments there are (num_segments - num_split_pos), elsewhere most of them
    full, live = int(rec["header"][3]) - int(rec["header"][4]), int(rec["header"][13])
    assert 0 < live <= full, (live, full)
    if which == "saturating":
        assert live < 0.6 * full, (live, full)
    if which == "init_opacity":
        assert live == full, (live, full)          # nothing saturates: every walk reaches the end of its list
    for k in ("color", "others", "radii", "n_contrib", "final_T"):
        assert torch.equal(rec[k], whole[k]), k
    # (paired workgroups -- the object scene's long tiles -- leave records whose sums are even + odd entries: 9e-6 seen)
    _grads_close(rec, whole, 1e-5 if int(rec["header"][17]) else 3e-6, which)


@pytest.mark.parametrize("split", ["0", "1", "1-relative"])
@pytest.mark.parametrize("which", ["uniform", "object_init", "object_opaque"])
def test_geom_blend_equals_the_full_blend(gpu_device, monkeypatch, which, split):
    """aux_planes = AUX_GEOM (planes 0-4: depth, alpha, normal -- the regularised Stage-3 regime with the upstream defaults
    lambda_dist = 0, depth_ratio = 0): colour and planes 0-4 are the full blend's, planes 5-7 zeros; the backward TAKES
    planes 5-7 as zero (they hold NaNs here) and gives what the full backward gives for zero-filled planes.  Unsplit:
    images bit-identical; split (the GEOM instance runs without the transmittance pre-pass: relative segments + repair of
    the saturating one): <= 2e-5 of scale, contributor counts identical.  "1": segment-parallel with the transmittance
    pre-pass (the default for this mode); "1-relative": without it (relative segments + repair of the saturating one,
    VIDU4D_SURFEL_SPEC_GEOM=1)."""
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    monkeypatch.setattr(_C, "_SPEC_GEOM", split == "1-relative")
    split = split[0]
    monkeypatch.setattr(_C, "_SPLIT", split)
    if which == "uniform":
        sc = make_scene(50_000, 256, 192, seed=41)
    elif which == "object_init":
        sc = make_object_scene(60_000, 256, radius=0.4, opacity_mode="init")     # long lists, nothing saturates
    else:
        sc = make_object_scene(60_000, 256, radius=0.4)                          # long lists, saturation inside them
    dc, do = (t.to(dev) for t in make_upstream_grads(sc.width, sc.height))
    do_zero = do.clone()
    do_zero[5:] = 0
    do_nan = do.clone()
    do_nan[5:] = float("nan")
    full = _run(sc, dev, dc, do_zero, aux=0)
    geom = _run(sc, dev, dc, do_nan, aux=_lib.AUX_GEOM)
    if split == "1":
        assert int(full["header"][3]) > 0 or which == "uniform"
    assert torch.equal(full["radii"], geom["radii"])
    assert torch.equal(full["n_contrib"][: sc.width * sc.height], geom["n_contrib"][: sc.width * sc.height]), "last contributor"
    assert float(geom["others"][5:].abs().max()) == 0.0
    assert float(full["others"][5].abs().max()) > 0.0
    if split == "0":
        assert torch.equal(full["color"], geom["color"])
        assert torch.equal(full["others"][:5], geom["others"][:5])
        _grads_close(geom, full, 2e-6, which)
    else:
        for a, b, what in ((geom["color"], full["color"], "colour"), (geom["others"][:5], full["others"][:5], "planes 0-4")):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), what
        _grads_close(geom, full, 2e-5, which)


def test_recorded_segments_of_stacked_frames(gpu_device):
    """Two stacked frames through dsr.rasterize_frames (what the bench and the trainer run): gradients with recorded
    segments against the whole-tile backward."""
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C, _lib
    from vidu4d_amd.synthetic import frame_motion
    dev = gpu_device
    W, H, N, F = 240, 176, 50_000, 2
    sc = make_scene(N, W, H, seed=7).to(dev)
    frames = [frame_motion(sc, 5 * f, 12) for f in range(F)]
    views = [dsr.GaussianRasterizationSettings(H, W, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                               sc.sh_degree, sc.campos, False, False)] * F
    dc, do = make_upstream_grads(W, H)
    dcs = torch.stack([(dc * (1 + 0.3 * f)).to(dev) for f in range(F)], 1)
    dos = torch.stack([(do * (1 - 0.2 * f)).to(dev) for f in range(F)], 1)
    M3, R4 = torch.stack([fr.means3D for fr in frames]), torch.stack([fr.rotations for fr in frames])

    def run(flags):
        leaves = [t.clone().requires_grad_(True) for t in (M3, torch.zeros_like(M3), sc.shs, sc.opacities, sc.scales, R4)]
        with _C.debug_flags(flags):
            out = dsr.rasterize_frames(*leaves, views)
            torch.autograd.backward([out[0], out[2]], [dcs, dos])
        return out, [t.grad for t in leaves]

    o1, g1 = run(0)
    o2, g2 = run(_lib.DEBUG_WHOLE_TILE_BACKWARD)
    assert torch.equal(o1[0], o2[0]) and torch.equal(o1[2], o2[2])
    for a, b, what in zip(g1, g2, ("means3D", "means2D", "sh", "opacity", "scales", "rotations")):
        assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max()), what
