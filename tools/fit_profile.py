import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
N, H, W, frames = int(__import__("os").environ.get("FIT_N", "200000")), 512, 512, 120
RADIUS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0  # object radius; camera at distance 3, tan(fov/2)=0.5
rng = np.random.default_rng(0)
import os
opts = dict(fg_motion="gs-bob", densify_until_iter=0, frame_streams=os.environ.get("FIT_STREAMS", "1") == "1")
opts.update(__import__("json").loads(os.environ.get("FIT_OPTS", "{}")))  # e.g. FIT_OPTS='{"canonical_params": false}' for an A/B
m = DeformableSurfels(opts, num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = RADIUS * pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32), )
with torch.no_grad(): m._scaling.add_(0.0)
tr = Stage3Trainer(m)
tr.current_steps = int(os.environ.get("FIT_STEP0", "0"))  # > 8000: normal-consistency regulariser on
m.pipeline.fused_post = os.environ.get("FIT_FUSED_POST", "1") == "1"
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(4)]
for b in batches: b["Kinv"] = batches[0]["Kinv"]  # one intrinsics tensor for the run (--force_center_cam)
for i in range(6): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
t0 = time.perf_counter(); K = int(os.environ.get("FIT_K", "10"))
for i in range(K): tr.train_step(batches[i % 4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"FIT_STEP 200k/512^2 radius {RADIUS}, 2 frames/step: {dt*1e3:.2f} ms/step = {2/dt:.1f} images/s")
if os.environ.get("FIT_PRINT_HINTS", "0") == "1":   # what the split decision sees
    from vidu4d_amd import _C
    print("FIT_HINTS deepest blended list position", dict(_C._depth_hint), "longest list", dict(_C._len_hint))
if os.environ.get("FIT_WALK_STATS", "0") == "1":   # what the backward's walk of one step looks like (vidu4d_surfel_blend_stats)
    from vidu4d_amd import _C
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    _C.count_next_walk(cnt)
    tr.train_step(batches[0]); torch.cuda.synchronize()
    c = [int(x) for x in cnt.tolist()]
    print(f"FIT_WALK entries staged {c[0]}, wave trips {c[1]} ({c[1] / max(1, c[0]):.2f} per entry), with a contributor {c[2]}, "
          f"contributing pairs {c[3]} = {c[3] / (64.0 * max(1, c[1])):.1%} of the lanes, lanes <=4/8/16/32/64: {c[5:10]}")
if os.environ.get("FIT_NO_TORCH_PROF", "0") == "1":
    sys.exit(0)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for i in range(3): tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
