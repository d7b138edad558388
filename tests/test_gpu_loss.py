"""csrc/loss.hip (five launches from the rasterizer's planes to the weighted loss terms and back) against the reference's
own numbers (tests/golden/refpy_losses.npz, written by the imported lab4d/engine/model.py) and against the torch
statement of the same arithmetic (stage3.compute_losses) on random frames with the learnable-background composite."""
import os

import numpy as np
import pytest
import torch

from tests.test_refpy_host import LOSS_CASES, close, load

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from vidu4d_amd.lab4d.deformable_surfels import _Args
    return _Args(dict(dict(lambda_normal=0.05, lambda_dist=100.0, lambda_dssim=0.0, rgb_wt=0.1, mask_wt=0.1), **kw))


def _planes(rendered, mask, dist):
    """(M,H,W,3) / (M,H,W,1) maps -> per-frame (3,H,W) colour and (8,H,W) auxiliary planes (alpha = plane 1, distortion = 6)."""
    M = rendered.shape[0]
    colors, allmaps = [], []
    for m in range(M):
        colors.append(rendered[m].permute(2, 0, 1).contiguous().requires_grad_(True))
        am = torch.zeros(8, *rendered.shape[1:3], device=rendered.device)
        am[1], am[6] = mask[m, ..., 0], dist[m, ..., 0]
        am[0], am[2:6], am[7] = 0.3, 0.7, 0.1  # (planes the loss must not read)
        allmaps.append(am.requires_grad_(True))
    return colors, allmaps


@pytest.mark.parametrize("through_total", [False, True])
@pytest.mark.parametrize("case", LOSS_CASES)
def test_fused_loss_matches_the_reference_numbers(gpu_device, case, through_total):
    from vidu4d_amd.lab4d.loss_fused import stage3_loss, unit_gradient
    dev = gpu_device
    r = load("refpy_losses.npz", dev)
    step = int(r[f"{case}_step"])
    colors, allmaps = _planes(r[f"{case}_in_rendered"], r[f"{case}_in_mask"], r[f"{case}_in_rend_dist"])
    batch = {"rgb": r[f"{case}_batch_rgb"], "mask": r[f"{case}_batch_mask"], "vis2d": r[f"{case}_batch_vis2d"],
             "is_detected": r[f"{case}_batch_is_detected"]}
    losses = stage3_loss(colors, allmaps, None, batch, step, _cfg())
    for k in ("rgb", "mask", "dist_loss"):
        close(losses[k], r[f"{case}_loss_{k}"], what=f"{case}:{k}", rtol=2e-5, atol=1e-8)
    terms = losses["rgb"] + losses["mask"] + losses["dist_loss"]
    assert float(losses["total"]) == float(terms) or (terms.isnan() and losses["total"].isnan())
    if through_total:  # the kernel's own sum, started from the cached unit gradient (what Stage3Trainer does)
        losses["total"].backward(gradient=unit_gradient(dev))
    else:
        terms.backward()
    g_r = torch.stack([c.grad.permute(1, 2, 0) for c in colors])
    g_m = torch.stack([a.grad[1][..., None] for a in allmaps])
    g_d = torch.stack([a.grad[6][..., None] for a in allmaps])
    close(g_r, r[f"{case}_g_rendered"], what="g_rendered", rtol=1e-4, atol=1e-9)
    close(g_m, r[f"{case}_g_mask"], what="g_mask", rtol=1e-4, atol=1e-9)
    close(g_d, r[f"{case}_g_rend_dist"], what="g_rend_dist", rtol=1e-4, atol=1e-9)
    for a in allmaps:
        assert float(a.grad[[0, 2, 3, 4, 5, 7]].abs().max()) == 0.0


@pytest.mark.parametrize("M,H,W,step,det", [(2, 64, 48, 100, None), (3, 33, 57, 9000, [1, 0, 1]), (1, 16, 16, 0, [0])])
def test_fused_loss_with_learnable_background_matches_torch(gpu_device, M, H, W, step, det):
    from vidu4d_amd.lab4d.loss_fused import stage3_loss
    from vidu4d_amd.lab4d.stage3 import compute_losses
    dev = gpu_device
    g = torch.Generator().manual_seed(M * 100 + H)
    rnd = lambda *s: torch.rand(*s, generator=g).to(dev)  # noqa: E731
    cfg = _cfg(lambda_dist=3.0, lambda_normal=0.0, lambda_dssim=0.2)
    batch = {"rgb": rnd(M, H, W, 3), "mask": (rnd(M, H, W, 1) > 0.6).float(), "vis2d": (rnd(M, H, W, 1) > 0.1).float()}
    if det is not None:
        batch["is_detected"] = torch.tensor(det, device=dev).bool()
    color0 = [rnd(3, H, W) for _ in range(M)]
    allmap0 = [rnd(8, H, W) for _ in range(M)]
    bg0 = rnd(3)
    res = {}
    for name in ("fused", "torch"):
        colors = [c.clone().requires_grad_(True) for c in color0]
        allmaps = [a.clone().requires_grad_(True) for a in allmap0]
        bg = bg0.clone().requires_grad_(True)
        if name == "fused":
            losses = stage3_loss(colors, allmaps, bg, batch, step, cfg)
        else:
            comp = [c + (1 - a[1:2]) * bg[:, None, None] for c, a in zip(colors, allmaps)]
            rendered = {"rendered": torch.stack([c.permute(1, 2, 0) for c in comp]),
                        "mask": torch.stack([a[1:2].permute(1, 2, 0) for a in allmaps]),
                        "rend_dist": torch.stack([a[6:7].permute(1, 2, 0) for a in allmaps])}
            losses = compute_losses(rendered, batch, step, cfg)
        total = losses["rgb"] * 1.5 + losses["mask"] * 0.5 + losses["dist_loss"]
        total.backward()
        res[name] = ([float(losses[k]) for k in ("rgb", "mask", "dist_loss")],
                     [c.grad.clone() for c in colors] + [a.grad.clone() for a in allmaps] + [bg.grad.clone()])
    # (a batch without any detected frame makes the balance weights 0/0: the silhouette term is NaN upstream, and here)
    assert np.allclose(res["fused"][0], res["torch"][0], rtol=2e-5, atol=1e-9, equal_nan=True), (res["fused"][0], res["torch"][0])
    for a, b in zip(res["fused"][1], res["torch"][1]):
        assert torch.equal(a.isnan(), b.isnan())
        a, b = torch.nan_to_num(a), torch.nan_to_num(b)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-12)


def test_trainer_step_stacked_frames_equal_per_frame_calls(gpu_device):
    """One Stage3Trainer step with the frames through one stacked launch set / one rasterizer call per frame (both with
    the fused loss): same losses, same gradients, same densification statistics."""
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device
    out = {}
    for flag in (True, False):
        torch.manual_seed(0)
        rng = np.random.default_rng(2)
        m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=10**6, stacked_frames=flag, frame_streams=False),
                              num_frames=8, device=dev)
        m.init_from_points(rng.normal(size=(4000, 3)).astype(np.float32) * 0.25, rng.uniform(size=(4000, 3)).astype(np.float32))
        tr = Stage3Trainer(m)
        batch = synthetic_batch(m, [1, 4, 6], 64, 48, seed=3)
        tr.bind_flat_gradients()
        losses = tr._forward_backward(batch, 10)
        stats = [(m._viewspace_points_batch[i].grad.clone(), m._radii_batch[i].clone(), m._visibility_filter_batch[i].clone())
                 for i in range(3)]
        out[flag] = ({k: float(v) for k, v in losses.items()}, tr._flat.clone(), stats)
    for k in ("rgb", "mask"):
        assert abs(out[True][0][k] - out[False][0][k]) <= 1e-6 * abs(out[False][0][k]) + 1e-9, k
    a, b = out[True][1], out[False][1]
    assert torch.allclose(a, b, rtol=1e-3, atol=2e-6 * float(b.abs().max()))
    for (ga, ra, va), (gb, rb, vb) in zip(out[True][2], out[False][2]):
        assert torch.equal(ra, rb) and torch.equal(va, vb)
        assert torch.allclose(ga, gb, rtol=1e-3, atol=2e-6 * float(gb.abs().max()))


def test_trainer_step_fused_loss_equals_torch_loss(gpu_device):
    """One Stage3Trainer step with the fused loss on / off: same loss values, same surfel gradients."""
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device
    out = {}
    for flag in (True, False):
        torch.manual_seed(0)
        rng = np.random.default_rng(2)
        m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0, fused_loss=flag, frame_streams=False),
                              num_frames=8, device=dev)
        m.init_from_points(rng.normal(size=(4000, 3)).astype(np.float32) * 0.25, rng.uniform(size=(4000, 3)).astype(np.float32))
        tr = Stage3Trainer(m)
        batch = synthetic_batch(m, [1, 4], 64, 64, seed=3)
        tr.bind_flat_gradients()
        losses = tr._forward_backward(batch, 10)
        out[flag] = ({k: float(v) for k, v in losses.items()}, tr._flat.clone())
    for k in ("rgb", "mask"):
        assert abs(out[True][0][k] - out[False][0][k]) <= 2e-5 * abs(out[False][0][k]) + 1e-9, k
    a, b = out[True][1], out[False][1]
    assert torch.allclose(a, b, rtol=1e-3, atol=2e-6 * float(b.abs().max()))
