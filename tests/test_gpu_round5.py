"""Round 5: re-entrancy of the host side.  The reference keeps all state in the three buffers a forward
returns (rasterize_points.cu:31-37, :92-103) and is re-entrant; here the little extra the host keeps -- hints, unchecked
deferred forwards, one-shot gradient outputs -- lives in a RasterContext the caller owns (vidu4d_amd/_C.py)."""
import threading

import numpy as np
import pytest
import torch

from tests.util import make_case, oracle_forward, to_np

pytestmark = pytest.mark.gpu


def _model(dev, n, seed, radius=0.25, **opts):
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    rng = np.random.default_rng(seed)
    o = dict(fg_motion="gs-bob", sh_degree=3, densify_until_iter=0)
    o.update(opts)
    torch.manual_seed(seed)
    m = DeformableSurfels(o, num_frames=8, device=dev)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    pts = radius * pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.8, 1.0, size=(n, 1)).astype(np.float32)
    m.init_from_points(pts, rng.uniform(size=(n, 3)).astype(np.float32))
    return m


def test_two_models_stepping_alternately_equal_each_alone(gpu_device):
    """Two DeformableSurfels of DIFFERENT sizes that render the SAME image size, their trainers stepping in turn: every
    model ends where it ends when it trains alone (to the noise of the float atomics), and neither sees the other's hints --
    with module-level hints keyed on the image shape the small model's pair count sized the large model's binning buffer,
    every alternation was an overflow + replay, and a model's deferred forwards sat in the other's `pending`."""
    from vidu4d_amd import _C
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev, H, W, steps = gpu_device, 96, 96, 4
    spec = {"big": dict(n=12000, seed=3, radius=0.3), "small": dict(n=1500, seed=4, radius=0.2)}

    def run(names):
        models = {k: _model(dev, **spec[k]) for k in names}
        trainers = {k: Stage3Trainer(models[k]) for k in names}
        batches = {k: [synthetic_batch(models[k], [2 * i, 2 * i + 1], H, W, seed=i) for i in range(steps)] for k in names}
        replays = {k: 0 for k in names}
        # count the replays (check_deferred returning False) per model
        _orig_check = _C.check_deferred

        def spy(context=None):
            ok = _orig_check(context)
            if not ok:
                for k in names:
                    if _C.current() is models[k].raster_context:
                        replays[k] += 1
            return ok
        _C.check_deferred = spy
        try:
            for i in range(steps):
                for k in names:
                    trainers[k].train_step(batches[k][i])
        finally:
            _C.check_deferred = _orig_check
        torch.cuda.synchronize(dev)
        return ({k: (models[k]._xyz.detach().clone(), models[k]._features_dc.detach().clone()) for k in names},
                {k: dict(models[k].raster_context.capacity_hint) for k in names}, replays, models)

    both, hints_both, replays_both, models = run(["big", "small"])
    assert not _C._pending and not _C._capacity_hint.get((W, H, str(dev), 2, None))   # nothing leaked into the default context
    for k in ("big", "small"):
        alone, hints_alone, replays_alone, _ = run([k])
        for a, b in zip(both[k], alone[k]):
            d = (a - b).abs()
            assert float(d.median()) <= 1e-6 and float(d.max()) <= 4 * 2.5e-3, (k, float(d.max()))
        # the model's own hints, the same as when it runs alone; no replay beyond the ones a lone run has (its first steps)
        assert set(hints_both[k]) == set(hints_alone[k]) and replays_both[k] == replays_alone[k], (k, replays_both, replays_alone)
        assert not models[k].raster_context.pending
    big_cap, small_cap = (max(hints_both[k].values()) for k in ("big", "small"))
    assert big_cap > 2 * small_cap, (big_cap, small_cap)


def _render(sc, dev, context=None):
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C
    d = sc.to(dev)
    rs = dsr.GaussianRasterizationSettings(d.height, d.width, d.tanfovx, d.tanfovy, d.bg, 1.0, d.viewmatrix, d.projmatrix,
                                           d.sh_degree, d.campos, False, False)
    leaves = [t.clone().requires_grad_(True) for t in (d.means3D, d.opacities, d.scales, d.rotations, d.shs)]
    m2d = torch.zeros_like(leaves[0], requires_grad=True)
    ctx = context if context is not None else _C.current()
    with ctx:
        color, radii, allmap = dsr.GaussianRasterizer(rs)(means3D=leaves[0], means2D=m2d, opacities=leaves[1], shs=leaves[4],
                                                          scales=leaves[2], rotations=leaves[3])
        (color.sum() + allmap[:5].sum()).backward()
    return color.detach(), allmap.detach(), radii, [t.grad for t in leaves]


def test_two_threads_rendering_concurrently(gpu_device):
    """Two Python threads, each on its own HIP stream, calling GaussianRasterizer.forward + backward in a loop on DIFFERENT
    scenes of the same image size, through the module-level API (each thread gets a default context of its own): every call
    returns what the same call returns single-threaded -- forward planes bit for bit (no atomics in the forward), gradients
    to the noise of the backward's float atomics."""
    dev = gpu_device
    scenes = {"a": make_case("small"), "b": make_case("small")}
    scenes["b"].means3D = scenes["b"].means3D * torch.tensor([0.6, 0.6, 1.0])    # a denser frame: other pair count, other hints
    want = {k: _render(sc, dev) for k, sc in scenes.items()}
    torch.cuda.synchronize(dev)
    errors, got = [], {k: [] for k in scenes}

    def worker(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(6):
                    got[k].append(_render(scenes[k], dev))
                stream.synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in scenes]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in scenes:
        wc, wa, wr, wg = want[k]
        st = oracle_forward(scenes[k])
        assert np.array_equal(to_np(wr), st["radii"])
        for c, a, r, g in got[k]:
            assert torch.equal(c, wc) and torch.equal(a, wa) and torch.equal(r, wr), k
            for x, y in zip(g, wg):
                scale = float(y.abs().max()) + 1e-30
                assert float((x - y).abs().max()) <= 2e-5 * scale, k


@pytest.mark.parametrize("aux", ["alpha", "geom"])
@pytest.mark.parametrize("opacity", [0.6, 0.05, 0.004])
def test_repairs_on_workgroups_of_their_own_equal_the_serial_combine(gpu_device, monkeypatch, opacity, aux):
    """The pre-pass-free segment-parallel forward (assume_unsaturated) adds its segments up in three launches since round 5
    -- scan, one workgroup per (tile, segment) some pixel saturates in, add up -- instead of one launch that walks a tile's
    saturating segments one after the other (VIDU4D_DEBUG_SERIAL_REPAIR).  Same operations in the same order: images,
    contributor counts, final transmittances and the per-segment state the backward starts from are BIT-identical; the
    gradients differ by the order of the backward's float atomics only.  0.6: every covered pixel saturates, in the first
    segments; 0.05: deep in the lists and not everywhere; 0.004: nothing saturates (no segment is flagged)."""
    from tests.test_gpu_parity import _concentrated_scene
    from vidu4d_amd import _C, _lib
    dev = gpu_device
    sc = _concentrated_scene(dev, opacity)
    dc, do = (t.to(dev) for t in __import__("vidu4d_amd.synthetic", fromlist=["x"]).make_upstream_grads(sc.width, sc.height))
    planes = _lib.AUX_ALPHA if aux == "alpha" else _lib.AUX_GEOM
    keep = [1] if aux == "alpha" else [0, 1, 2, 3, 4]
    do_live = torch.zeros_like(do)
    do_live[keep] = do[keep]
    monkeypatch.setattr(_C, "_SPLIT", "1")
    monkeypatch.setattr(_C, "_SPEC", True)
    monkeypatch.setattr(_C, "_SPEC_GEOM", True)
    e = torch.empty(0, device=dev)

    def run(flags):
        with _C.debug_flags(flags):
            out = _C.rasterize_gaussians(sc.bg, sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, sc.viewmatrix,
                                         sc.projmatrix, sc.tanfovx, sc.tanfovy, sc.height, sc.width, sc.shs, 3, sc.campos,
                                         False, False, aux_planes=planes)
            R, color, others, radii, geom, binning, img = out
            g = _C.rasterize_gaussians_backward(sc.bg, sc.means3D, radii, e, sc.scales, sc.rotations, 1.0, e, sc.viewmatrix,
                                                sc.projmatrix, sc.tanfovx, sc.tanfovy, dc, do_live, sc.shs, 3, sc.campos, geom,
                                                R, binning, img, False, aux_planes=planes)
        n = sc.width * sc.height
        ncon = _C.read_state("n_contrib", None, geom, binning, img, sc.num_surfels, sc.width, sc.height, torch.int32, 2 * n)
        fT = _C.read_state("final_T", None, geom, binning, img, sc.num_surfels, sc.width, sc.height, torch.float32, 3 * n)
        return color, others, ncon, fT, [t for t in g if t.numel()], geom[:64].view(torch.int32).cpu()
    serial = run(_lib.DEBUG_SERIAL_REPAIR)
    three = run(0)
    assert int(serial[5][3]) > 0 and int(serial[5][5]) == 1, "the scene must be blended segment-parallel"   # num_segments, split_used
    assert int(three[5][6]) == 0 and int(serial[5][6]) == 0                                                   # truncated
    assert torch.equal(three[5][8:9], serial[5][8:9])                                                          # min final T, bits
    for a, b, what in zip(three[:4], serial[:4], ("colour", "planes", "contributors", "final T")):
        assert torch.equal(a, b), what
    for i, (a, b) in enumerate(zip(three[4], serial[4])):
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max()) + 1e-12, i
