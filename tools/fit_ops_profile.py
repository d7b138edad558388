import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
N, H, W, frames = 200000, 512, 512, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m)
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(4)]
for i in range(6): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    for i in range(2): tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:45]:
    print(f"{e.key:32s} n={e.count/2:5.1f} cuda_us/step={e.device_time_total/2:8.1f} shapes={str(e.input_shapes)[:90]}")

print("---- by call site")
rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:70]:
    site = next((f for f in e.stack if "vidu4d_amd" in f and "profiler" not in f), e.stack[0] if e.stack else "?")
    print(f"{e.key:28s} n={e.count/2:5.1f} cuda_us/step={e.device_time_total/2:7.1f}  {site.replace('/root/repo/', '')[-90:]}")
