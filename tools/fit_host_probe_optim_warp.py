"""Host cost of the fitting step with TRAINING networks: tiny scene, cProfile.  Usage (GPU box): python tools/fit_host_probe_optim_warp.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
N, H, W, frames = 2000, 64, 64, 120
rng = np.random.default_rng(0)
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m, m.opts | dict(gs_optim_warp=True)); tr.current_steps = 12001
m.active_sh_degree = m.max_sh_degree
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(8)]
for b in batches: b["Kinv"] = batches[0]["Kinv"]
for i in range(20): tr.train_step(batches[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100): tr.train_step(batches[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"FIT_HOST optim_warp: {dt*1e3:.3f} ms per step with kernels of next to no duration")
pr = cProfile.Profile(); pr.enable()
for i in range(50): tr.train_step(batches[i % 8])
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(60)
