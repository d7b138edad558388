"""Where the fitting step with TRAINING networks (--gs_optim_warp=True) spends its time: torch profiler tables (GPU box)."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
N, H, W, frames = 200000, 512, 512, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0, fused_warp_trainable=os.environ.get("FUSED", "1") == "1",
                           captured_step=os.environ.get("CAPTURED", "0") == "1"), num_frames=frames, device=dev)   # (eager: the ops keep their names)
d = rng.normal(size=(N, 3)).astype(np.float32)
pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 1.0, size=(N, 1)).astype(np.float32) ** (1 / 3)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m, m.opts | dict(gs_optim_warp=True))
m.active_sh_degree = m.max_sh_degree
tr.current_steps = 12001
batches = [synthetic_batch(m, [(2 * i) % frames, (2 * i + 1) % frames], H, W, seed=i) for i in range(4)]
for i in range(8): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): tr.train_step(batches[i % 4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"OPTIM_WARP step {dt*1e3:.2f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(3): tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=90))
if os.environ.get("BY_STACK", "0") == "1":   # which lines of the host code the small launches come from
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof2:
        for i in range(3): tr.train_step(batches[i % 4])
        torch.cuda.synchronize()
    print(prof2.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=60, max_src_column_width=110))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=20, max_name_column_width=70))
