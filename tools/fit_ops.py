"""Launch census of one Stage-3 step (which torch ops the ~600 launches come from)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device("cuda:0")
N, H, W, frames = 200_000, 512, 512, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m)
b = synthetic_batch(m, [0, 1], H, W)
for i in range(3): tr.train_step(b)
torch.cuda.synchronize()
import vidu4d_amd.lab4d.stage3 as S
orig_losses = S.compute_losses
def wrapped(*a, **k):
    with record_function("SECTION_losses"):
        return orig_losses(*a, **k)
S.compute_losses = wrapped
orig_rf = m.render_frames
def rf(*a, **k):
    with record_function("SECTION_render_frames"):
        return orig_rf(*a, **k)
m.render_frames = rf
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.train_step(b)
    torch.cuda.synchronize()
ev = prof.key_averages()
tot = sum(e.count for e in ev if e.key == "hipLaunchKernel")
print("launches per step:", tot)
rows = sorted([(e.key, e.count, e.cpu_time_total, e.self_cpu_time_total) for e in ev], key=lambda r: -r[2])
for r in rows[:45]:
    print(f"{r[0][:70]:70s} n={r[1]:4d} cpu_total={r[2]/1e3:7.2f}ms self={r[3]/1e3:7.2f}ms")
