"""Which small torch ops a fitting step with TRAINING networks still issues, by input shape (the eager loop: ops keep their names)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
N, H, W, frames = 200000, 512, 512, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0, captured_step=False, graphed_warp_networks="inline"), num_frames=frames, device=dev)
d = rng.normal(size=(N, 3)).astype(np.float32)
pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 1.0, size=(N, 1)).astype(np.float32) ** (1 / 3)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m, m.opts | dict(gs_optim_warp=True))
m.active_sh_degree = m.max_sh_degree
tr.current_steps = 12001
batches = [synthetic_batch(m, [(2 * i) % frames, (2 * i + 1) % frames], H, W, seed=i) for i in range(4)]
for i in range(8): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    for i in range(4): tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print("per step: calls, self GPU us, op, input shapes")
for e in rows[:70]:
    print(f"{e.count / 4:6.2f} {e.self_device_time_total / 4:8.1f}  {e.key:28s} {str(e.input_shapes)[:150]}")
