"""Round 6: the XCD-local longest-first schedule (csrc/binning.hip grouped_order) changes WHERE and WHEN a tile
is blended, never what comes out: forward planes / integers bit-identical with the one-queue schedule of rounds 1-5, gradients
to the order of the backward's float atomics.  The reference walks tiles in blockIdx order (forward.cu:265-280,
backward.cu:143-160); any order is a valid one."""
import numpy as np
import pytest
import torch

from tests.util import make_case, to_np
from vidu4d_amd.synthetic import make_scene, make_upstream_grads

pytestmark = pytest.mark.gpu


def _run(sc, dev, block, stacked=0, split=None, aux=0):
    """forward + backward under a context of its own with the schedule's block size; -> planes, integers, grads, state"""
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C
    d = sc.to(dev)
    ctx = _C.RasterContext()
    ctx.xcd_block = block
    rs = dsr.GaussianRasterizationSettings(d.height, d.width, d.tanfovx, d.tanfovy, d.bg, 1.0, d.viewmatrix, d.projmatrix,
                                           d.sh_degree, d.campos, False, False)
    leaves = [t.clone().requires_grad_(True) for t in (d.means3D, d.opacities, d.scales, d.rotations, d.shs)]
    dc, do = make_upstream_grads(sc.width, sc.height)
    dc, do = dc.to(dev), do.to(dev)
    out = {}
    with ctx:
        if stacked:
            F = stacked
            means = torch.stack([leaves[0] * (1.0 + 0.02 * f) for f in range(F)])
            rots = torch.stack([leaves[3]] * F)
            m2d = torch.zeros_like(means, requires_grad=True)
            color, radii, allmap = dsr.rasterize_frames(means, m2d, leaves[4], leaves[1], leaves[2], rots, [rs] * F, aux_planes=aux)
            gc = torch.stack([dc] * F, 1).contiguous()
            go = torch.stack([do] * F, 1).contiguous()
        else:
            m2d = torch.zeros_like(leaves[0], requires_grad=True)
            color, radii, allmap = dsr.GaussianRasterizer(rs)(means3D=leaves[0], means2D=m2d, opacities=leaves[1], shs=leaves[4],
                                                              scales=leaves[2], rotations=leaves[3])
            gc, go = dc, do
        saved = color.grad_fn.saved_tensors   # (before the backward frees them)
        torch.autograd.backward([color, allmap], [gc, go])
    torch.cuda.synchronize(dev)
    out["color"], out["allmap"], out["radii"] = color.detach(), allmap.detach(), radii
    out["grads"] = [t.grad for t in leaves] + [m2d.grad]
    out["bufs"] = [t for t in saved if t.dtype == torch.uint8]   # geom, binning, img in the order they were saved
    return out


def _state(o, sc, what, dtype, count, frames=1):
    from vidu4d_amd import _C
    geom, binning, img = o["bufs"]
    return _C.read_state(what, {}, geom, binning, img, sc.means3D.shape[0], sc.width, sc.height, dtype, count, frames=frames).numpy()


def _group(t, gx, frame_tiles, B):
    frame, r = t // frame_tiles, t % frame_tiles
    return ((r % gx) // B + 3 * ((r // gx) // B) + 5 * frame) & 7


@pytest.mark.parametrize("block", [1, 2, 4])
@pytest.mark.parametrize("case", ["small", "ragged", "huge"])
def test_xcd_local_schedule_changes_no_result(gpu_device, case, block):
    sc = make_case(case)
    a, b = _run(sc, gpu_device, 0), _run(sc, gpu_device, block)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["allmap"], b["allmap"]) and torch.equal(a["radii"], b["radii"])
    for name, dt, n in (("n_contrib", torch.int32, 2 * sc.width * sc.height), ("final_T", torch.float32, 3 * sc.width * sc.height)):
        assert np.array_equal(_state(a, sc, name, dt, n), _state(b, sc, name, dt, n)), name
    for ga, gb in zip(a["grads"], b["grads"]):
        scale = float(ga.abs().max()) + 1e-30
        assert float((ga - gb).abs().max()) <= 2e-5 * scale   # (the float atomics' order; the same kernel twice differs by 5e-6)


@pytest.mark.parametrize("stacked", [0, 2])
def test_xcd_local_schedule_properties(gpu_device, stacked):
    """What the schedule promises (binning.hip grouped_order): a permutation of the tiles; the split tiles -- those the
    segment table lists -- exactly the first positions; inside a group longest first; position mod 8 = the tile's group
    wherever all eight queues still hold a tile; the same for the tails."""
    B = 2
    sc = make_scene(60000, 256, 256, seed=3)   # 16 x 16 tiles per frame, lists of ~700 entries: recorded segments on
    o = _run(sc, gpu_device, B, stacked=stacked)
    F = max(stacked, 1)
    gx = (sc.width + 15) // 16
    frame_tiles = gx * ((sc.height + 15) // 16)
    T = frame_tiles * F
    hdr = _state(o, sc, "header", torch.int32, 64, frames=F)
    order = _state(o, sc, "tile_order", torch.int32, T, frames=F)
    tails = _state(o, sc, "tail_order", torch.int32, T, frames=F)
    ranges = _state(o, sc, "ranges", torch.int32, 2 * T, frames=F).reshape(T, 2)
    lens = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
    assert hdr[15] == B and hdr[5] == 2   # (xcd_block; split_used: recorded segments)
    assert sorted(order.tolist()) == list(range(T)) and sorted(tails.tolist()) == list(range(T))
    max_len, S = int(hdr[2]), int(hdr[4])
    cls = 1023 - (lens << 10) // (max_len + 1)
    cls_min = 1023 - (min(320, max_len) << 10) // (max_len + 1)
    split = cls < cls_min
    assert S == int(split.sum()) and split[order[:S]].all() and not split[order[S:]].any()
    grp = np.array([_group(t, gx, frame_tiles, B) for t in range(T)])
    for g in range(8):   # longest first inside every queue (by length class, as the one-queue schedule)
        for region in (order[:S], order[S:]):
            c = cls[region][grp[region] == g]
            assert (np.diff(c) >= 0).all(), g
    na = np.array([int((split & (grp == g)).sum()) for g in range(8)])
    nb = np.array([int((~split & (grp == g)).sum()) for g in range(8)])
    pa = np.arange(8 * na.min())
    assert (grp[order[pa]] == pa % 8).all()
    pb = S + np.arange(8 * nb.min())
    assert (grp[order[pb]] == pb % 8).all()
    assert 8 * na.min() >= 0.8 * S   # (the groups are balanced: most of the schedule is aligned)
    nt = np.bincount(grp, minlength=8)
    pt = np.arange(8 * nt.min())
    assert (grp[tails[pt]] == pt % 8).all()
    assert hdr[16] == 1 and hdr[13] % 8 == 0   # (live_xcd; the padded count of live full segments)


def test_xcd_local_schedule_stacked_and_modes(gpu_device):
    """Stacked frames, the colour / planes-0-4 instances and a forced segment-parallel forward under the XCD-local schedule."""
    from vidu4d_amd import _C
    from vidu4d_amd.diff_surfel_rasterization import AUX_ALPHA, AUX_GEOM
    sc = make_scene(30000, 160, 128, seed=9)
    for aux in (0, AUX_ALPHA, AUX_GEOM):
        a, b = _run(sc, gpu_device, 0, stacked=2, aux=aux), _run(sc, gpu_device, 2, stacked=2, aux=aux)
        assert torch.equal(a["color"], b["color"]) and torch.equal(a["allmap"], b["allmap"]) and torch.equal(a["radii"], b["radii"])
        for ga, gb in zip(a["grads"], b["grads"]):
            assert float((ga - gb).abs().max()) <= 2e-5 * (float(ga.abs().max()) + 1e-30)
    old = _C._SPLIT
    try:
        _C._SPLIT = "1"
        dense = make_scene(40000, 96, 96, seed=4, sigma_px=3.0)   # lists beyond 1024 entries: split tiles
        a, b = _run(dense, gpu_device, 0), _run(dense, gpu_device, 2)
        assert torch.equal(a["radii"], b["radii"])
        for x, y in ((a["color"], b["color"]), (a["allmap"], b["allmap"])):
            assert float((x - y).abs().max()) <= 2e-6 * (float(x.abs().max()) + 1e-30)
        for ga, gb in zip(a["grads"], b["grads"]):
            assert float((ga - gb).abs().max()) <= 2e-5 * (float(ga.abs().max()) + 1e-30)
    finally:
        _C._SPLIT = old
