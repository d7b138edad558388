"""Same-process A/B of the fitting step with TRAINING networks (--gs_optim_warp=True): every variant switches ONE of round 5's
changes off (options of DeformableSurfels / Stage3Trainer), interleaved repeats, median ms per step.  GPU box.
200 000 surfels, 512 x 512, 2 frames per step, step 12 001 (AdamW stepping)."""
import os, sys, time, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
from vidu4d_amd.lab4d import bob_warp

dev = torch.device("cuda:0")
N, H, W, frames = 200000, 512, 512, 120
VARIANTS = [
    ("all on", {}),
    ("delta MLP as library GEMMs", dict(fused_skin_field_trainable=False)),
    ("dense stacks as library calls", dict(fused_dense_stacks=False)),
    ("one branch instead of three", dict(parallel_network_branches=False)),
    ("quaternion algebra as torch ops", dict(fused_bone_tables=False)),
    ("torch.cuda.make_graphed_callables", dict(graphed_warp_networks="torch")),
    ("networks eager (no graphs)", dict(graphed_warp_networks=False)),
    ("AdamW groups per tensor", dict(network_param_groups="per_tensor")),
    ("un-fused torch warp (rounds 1-4)", dict(fused_warp_trainable=False)),
    ("all on, step 5 000 (before AdamW: gradients accumulate)", dict(_step=5000)),
    ("un-fused torch warp, step 5 000", dict(fused_warp_trainable=False, _step=5000)),
]
VARIANTS.insert(2, ("... and their weight gradients as plain GEMMs", dict(fused_skin_field_trainable=False, _split_k=0)))


def make(opts):
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    o = dict(fg_motion="gs-bob", densify_until_iter=0)
    o.update({k: v for k, v in opts.items() if not k.startswith("_")})
    m = DeformableSurfels(o, num_frames=frames, device=dev)
    d = rng.normal(size=(N, 3)).astype(np.float32)
    pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 1.0, size=(N, 1)).astype(np.float32) ** (1 / 3)
    m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
    tr = Stage3Trainer(m, m.opts | dict(gs_optim_warp=True, num_rounds=120, iters_per_round=200))
    m.active_sh_degree = m.max_sh_degree
    tr.current_steps = opts.get("_step", 12001)
    return m, tr


def run(tr, batches, steps, split_k):
    old = bob_warp.SPLIT_K_CHUNK
    # (the chunk is a default argument of delta_raw_T / feature_major_linear: patch the functions' defaults for the variant)
    if split_k is not None:
        bob_warp.SkinningField.delta_raw_T.__defaults__ = (split_k,)
        bob_warp.feature_major_linear.__defaults__ = (split_k,)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train_step(batches[i % len(batches)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    finally:
        bob_warp.SkinningField.delta_raw_T.__defaults__ = (old,)
        bob_warp.feature_major_linear.__defaults__ = (old,)


models = []
for name, opts in VARIANTS:
    m, tr = make(opts)
    batches = [synthetic_batch(m, [(2 * i) % frames, (2 * i + 1) % frames], H, W, seed=i) for i in range(4)]
    sk = opts.get("_split_k")
    run(tr, batches, 8, sk)   # warm-up: capacity hints, graph capture
    models.append((name, tr, batches, sk, []))
for rep in range(7):
    for name, tr, batches, sk, times in models:
        times.append(run(tr, batches, 30, sk))
base = statistics.median(models[0][4])
print(f"{'variant':58s} ms/step (median of 7 x 30 steps)   images/s   vs all-on")
for name, _, _, _, times in models:
    med = statistics.median(times)
    print(f"{name:58s} {med:7.3f}  [{min(times):.3f} .. {max(times):.3f}]   {2e3 / med:8.1f}   {med / base:5.2f} x")
