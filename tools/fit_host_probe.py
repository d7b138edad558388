"""The HOST's cost of a fitting step: the trainer on a scene whose kernels take next to nothing (2 000 surfels, 64^2), and a
cProfile of where it goes.  Usage (GPU box): python tools/fit_host_probe.py [step0]"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
dev = torch.device("cuda:0")
STEP0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, H, W, frames = 2000, 64, 64, 120
rng = np.random.default_rng(0)
# CAPTURED=1 (default): the plain steps replayed from one captured hipGraph (lab4d/captured_step.py; forced on -- with frozen
# networks the trainer's "auto" keeps the eager loop, which is GPU-bound at real sizes); CAPTURED=0: the eager loop
CAPTURED = os.environ.get("CAPTURED", "1") == "1"
m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0, captured_step=CAPTURED), num_frames=frames, device=dev)
pts = rng.normal(size=(N, 3)).astype(np.float32); pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 1.0, size=(N, 1)).astype(np.float32)
m.init_from_points(pts, rng.uniform(size=(N, 3)).astype(np.float32))
tr = Stage3Trainer(m); tr.current_steps = STEP0
if STEP0: m.active_sh_degree = m.max_sh_degree
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(8)]
for b in batches: b["Kinv"] = batches[0]["Kinv"]
for i in range(40): tr.train_step(batches[i % 8])
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(400): tr.train_step(batches[i % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
print(f"FIT_HOST step0={STEP0} captured_step={CAPTURED}: {dt*1e3:.3f} ms per step (2 frames) with kernels of next to no duration "
      f"(captured: part of it is the host WAITING for the previous step's header -- see `verdict` in the profile below); {tr.captured_stats}")
pr = cProfile.Profile(); pr.enable()
for i in range(200): tr.train_step(batches[i % 8])
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
