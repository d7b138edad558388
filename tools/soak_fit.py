"""Soak run of the Stage-3 loop (teacher -> student, 16 frames, 192^2): densify / prune every 100 steps from step 200,
opacity reset at 600, SH degree step at 1000, frozen warp / camera networks (the fused HIP warp), once on the trainer's
default path (canonical parameters, alpha-only blend, stacked frames, fused loss) and once with those extensions off.
Prints surfel counts, the loss trajectory of both runs and the final image error against the teacher; every loss and
parameter must stay finite.  Usage (GPU box): python tools/soak_fit.py [steps]
SOAK_STEP0=7700 SOAK_DENSIFY_UNTIL=9000 SOAK_OPACITY_RESET=3000: the run crosses step 8000, where the normal-consistency
regulariser switches on (model.py:817-842), with densification still going (round 3: profiles/r03_soak_fit_geometry.txt)."""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, make_intrinsics_inv

dev = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
TRAIN_NETS = os.environ.get("SOAK_TRAIN_NETS", "0") == "1"   # the warp / camera / skinning networks train (round 5)
H = W = 192
FRAMES = 16


def model(n, seed, **opts):
    rng = np.random.default_rng(seed)
    o = dict(fg_motion="gs-bob", sh_degree=3)
    o.update(opts)
    torch.manual_seed(seed)
    m = DeformableSurfels(o, num_frames=FRAMES, device=dev)
    pts = rng.normal(size=(n, 3)).astype(np.float32)
    pts = 0.3 * pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.7, 1.0, size=(n, 1)).astype(np.float32)
    m.init_from_points(pts, rng.uniform(size=(n, 3)).astype(np.float32))
    return m


teacher = model(20000, 1, gs_learnable_bg=False)
with torch.no_grad():
    teacher._opacity.fill_(2.0)
    teacher._features_dc.normal_(0.0, 1.0)
    teacher._features_rest.normal_(0.0, 0.1)
fid = torch.arange(FRAMES, device=dev)
Kinv = make_intrinsics_inv(FRAMES, H, W, device=dev)
with torch.no_grad():
    tgt = teacher.render_frames(fid, Kinv, [H] * FRAMES, [W] * FRAMES)
nets = {k: v for k, v in teacher.state_dict().items() if k.startswith(("warp.", "camera_mlp."))}


def run(tag, **opts):
    s = model(8000, 2, densify_from_iter=200, densification_interval=100, densify_until_iter=int(os.environ.get("SOAK_DENSIFY_UNTIL", "1200")),
              opacity_reset_interval=int(os.environ.get("SOAK_OPACITY_RESET", "600")),
              densify_grad_threshold=5e-6,
              # (the radius-outlier pass of step 8000, 10000, ... keeps points with > 20 neighbours within 0.004 --
              # trainer.py:573-588, a fixed radius -- which on this sparse toy cloud is nobody: switched off here)
              outlier_filtering_interval=10 ** 9, **opts)
    s.load_state_dict(nets, strict=False)
    if TRAIN_NETS:   # --gs_optim_warp=True, the reference's default: gradients accumulate from step 0, AdamW from step 300
        tr = Stage3Trainer(s, s.opts | dict(gs_optim_warp=True, optim_warp_neus_iters=300, num_rounds=STEPS // 200 + 2,
                                            iters_per_round=200))
        assert tr.optim_warp and s.warp_networks_train()
    else:
        for mod in (s.warp, s.camera_mlp):
            for p in mod.parameters():
                p.requires_grad_(False)
        tr = Stage3Trainer(s)
    tr.current_steps = int(os.environ.get("SOAK_STEP0", "0"))  # > 8000: the normal-consistency regulariser is on
    hist, counts = [], []
    t0 = time.perf_counter()
    for step in range(STEPS):
        ids = [(2 * step) % FRAMES, (2 * step + 1) % FRAMES]
        batch = {"frameid": torch.tensor(ids, device=dev), "Kinv": make_intrinsics_inv(2, H, W, device="cpu"),
                 "H": [H, H], "W": [W, W], "rgb": tgt["rendered"][ids], "mask": tgt["mask"][ids].detach(),
                 "vis2d": torch.ones(2, H, W, 1, device=dev)}
        losses = tr.train_step(batch)
        if step % 25 == 0 or step == STEPS - 1:
            v = float(sum(losses.values()))
            assert np.isfinite(v), (tag, step, losses)
            hist.append(round(v, 5))
            counts.append(int(s._xyz.shape[0]))
    tr.settle()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{tag}] captured-step statistics: {tr.captured_stats}")
    for n, p in s.named_parameters():
        assert torch.isfinite(p).all(), (tag, n)
    with torch.no_grad():
        out = s.render_frames(fid, Kinv, [H] * FRAMES, [W] * FRAMES)
    err = float((out["rendered"] - tgt["rendered"]).abs().mean())
    merr = float((out["mask"] - tgt["mask"]).abs().mean())
    print(f"[{tag}] {STEPS} steps in {dt:.1f} s ({1e3 * dt / STEPS:.2f} ms/step); surfels {counts[0]} -> {counts[-1]} "
          f"(max {max(counts)}); SH degree {s.active_sh_degree}; loss {hist[0]} -> {hist[-1]}; "
          f"mean |image error| {err:.4f}, mean |mask error| {merr:.4f}")
    print(f"[{tag}] loss every 25 steps:", hist)
    return hist, err


if os.environ.get("SOAK_CAPTURED_AB", "0") == "1":
    # round 6: the plain steps replayed from ONE captured hipGraph (the trainer's default) against the eager loop of rounds 1-5
    h1, e1 = run("captured steps (default)" + (", networks train" if TRAIN_NETS else ""))
    h2, e2 = run("eager loop (captured_step=False)" + (", networks train" if TRAIN_NETS else ""), captured_step=False)
    worst = max(abs(a - b) / max(abs(b), 1e-9) for a, b in zip(h1, h2))
    print(f"captured vs eager: largest relative difference of the loss samples {worst:.2e}")
elif TRAIN_NETS:
    h1, e1 = run("networks train, default path (captured graphs, fused kernels)")
    h2, e2 = run("networks train, un-fused torch warp (rounds 1-4)", fused_warp_trainable=False)
else:
    h1, e1 = run("default path")
    h2, e2 = run("extensions off", canonical_params=False, alpha_only_blend=False, stacked_frames=False, fused_loss=False)
# (once the normal-consistency term is on, the loss carries its constant lambda * (1 - ...) ~ 0.049 on a mostly empty image:
# convergence is judged on the image error then)
if int(os.environ.get("SOAK_STEP0", "0")) + STEPS <= 8000:
    assert h1[-1] < 0.5 * h1[0] and h2[-1] < 0.5 * h2[0], "the fit did not converge"
assert e1 < 0.01 and e2 < 0.01, (e1, e2)
assert abs(e1 - e2) < 0.25 * max(e1, e2) + 0.01, (e1, e2)
print("SOAK OK")
