"""Same-process A/B of debug switches on the dense Stage-3 fitting scene: stage timers (HIP events on the launch stream)
of the blend kernels, the flag flipped every 15 steps, 6 rounds.  Usage: python tools/split_order_ab.py <radius> <step0> <flagA> <flagB>"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd import _C, _lib
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
RADIUS, STEP0 = float(sys.argv[1]), int(sys.argv[2])
FLAGS = [int(sys.argv[3]), int(sys.argv[4])]
N, H, W, frames = 200000, 512, 512, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0, frame_streams=True), num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = RADIUS * pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m); tr.current_steps = STEP0
if STEP0: m.active_sh_degree = m.max_sh_degree
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(4)]
for b in batches: b["Kinv"] = batches[0]["Kinv"]
for i in range(12): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
res = {f: [] for f in FLAGS}
for rnd in range(6):
    for f in FLAGS:
        m.raster_context.debug_flags = f
        for i in range(3): tr.train_step(batches[i % 4])
        torch.cuda.synchronize(); _lib.profile_read(reset=True); _lib.profile_enable(True)
        import time; t0 = time.perf_counter()
        for i in range(15): tr.train_step(batches[i % 4])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
        _lib.profile_enable(False)
        p = _lib.profile_read(reset=True)
        res[f].append((dt * 1e3, {k: round(ms / max(1, n), 4) for k, (ms, n) in p.items() if "blend" in k}))
for f in FLAGS:
    steps = [r[0] for r in res[f]]
    keys = res[f][0][1].keys()
    print(f"flags {f}: step ms median {np.median(steps):.3f} (min {min(steps):.3f});", {k: round(float(np.median([r[1][k] for r in res[f]])), 4) for k in keys})
