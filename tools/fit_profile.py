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
with torch.no_grad(): m._scaling.add_(0.0); m._opacity.add_(float(os.environ.get('FIT_OPACITY_SHIFT', '0')))   # (logits: +3 turns the initial 0.1 into 0.69 -- a trained, saturating cloud)
if os.environ.get("FIT_OPTIM_WARP", "0") == "1":   # networks that train (--gs_optim_warp=True); FIT_STEP0=12001: AdamW stepping
    tr = Stage3Trainer(m, m.opts | dict(gs_optim_warp=True, num_rounds=120, iters_per_round=200))
else:
    tr = Stage3Trainer(m)
tr.current_steps = int(os.environ.get("FIT_STEP0", "0"))  # > 8000: normal-consistency regulariser on
m.pipeline.fused_post = os.environ.get("FIT_FUSED_POST", "1") == "1"
batches = [synthetic_batch(m, [(2*i) % frames, (2*i+1) % frames], H, W, seed=i) for i in range(4)]
for b in batches: b["Kinv"] = batches[0]["Kinv"]  # one intrinsics tensor for the run (--force_center_cam)
if os.environ.get("FIT_PRINT_HINTS", "0") == "1":   # (the split decisions of the run, as taken)
    from vidu4d_amd import _C as _Cm
    _decisions = []
    _orig_auto = _Cm.auto_split

    def _spy_auto(depth, long_tiles, cus):
        r = _orig_auto(depth, long_tiles, cus)
        _decisions.append((depth, long_tiles, bool(r)))
        return r
    _Cm.auto_split = _spy_auto
for i in range(6): tr.train_step(batches[i % 4])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
t0 = time.perf_counter(); K = int(os.environ.get("FIT_K", "10"))
for i in range(K): tr.train_step(batches[i % 4])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"FIT_STEP 200k/512^2 radius {RADIUS}, 2 frames/step: {dt*1e3:.2f} ms/step = {2/dt:.1f} images/s")
if os.environ.get("FIT_TILE_STATS", "0") == "1":   # how far the forward's walk of each tile has to go (last contributor per tile)
    from vidu4d_amd import _C
    seen = {}
    orig = _C.rasterize_gaussians

    def spy(*a, **k):
        out = orig(*a, **k)
        seen["out"], seen["P"], seen["F"] = out, a[1].shape[-2] if a[1].dim() == 3 else a[1].shape[0], (a[1].shape[0] if a[1].dim() == 3 else 1)
        return out
    _C.rasterize_gaussians = spy
    tr.train_step(batches[0]); torch.cuda.synchronize()
    _C.rasterize_gaussians = orig
    R, color, others, radii, geom, binning, img = seen["out"]
    F = seen["F"]
    T = F * (H // 16) * (W // 16)
    rng_ = _C.read_state("ranges", None, geom, binning, img, seen["P"], W, H, torch.int32, 2 * T, frames=F).numpy().reshape(-1, 2)
    nc = _C.read_state("n_contrib", None, geom, binning, img, seen["P"], W, H, torch.int32, 2 * F * H * W, frames=F).numpy()
    last = nc[: F * H * W].reshape(F, H // 16, 16, W // 16, 16).transpose(0, 1, 3, 2, 4).reshape(-1, 256)
    lens = (rng_[:, 1] - rng_[:, 0]).astype(np.int64)
    walk = last.max(axis=1).astype(np.int64)              # entries the tile's walk covers
    alive_at = lambda k: (last > k).sum(axis=1)           # pixels still blending beyond entry k
    long_ = lens > 320
    print(f"FIT_TILES {T} tiles, {int(long_.sum())} longer than 320; list length median / max of those {int(np.median(lens[long_]))} / {int(lens.max())}; "
          f"walk length (deepest pixel) median / p90 / max {int(np.median(walk[long_]))} / {int(np.percentile(walk[long_], 90))} / {int(walk.max())}; "
          f"sum of walks {int(walk.sum())}")
    for k in (1024, 1536, 2048, 3072):
        u = walk > k
        print(f"   beyond entry {k}: {int(u.sum())} tiles still walking, {int((walk[u] - k).sum())} entries left in total, "
              f"alive pixels per such tile median {int(np.median(alive_at(k)[u])) if u.any() else 0}")
if os.environ.get("FIT_PRINT_HINTS", "0") == "1":   # what the split decision sees
    from vidu4d_amd import _C
    print("FIT_DECISIONS first", _decisions[:8], "last", _decisions[-4:], "split in", sum(d[2] for d in _decisions), "of", len(_decisions))
    rc = m.raster_context
    print("FIT_HINTS deepest blended list position", list(rc.depth_hint.values()), "longest list", list(rc.len_hint.values()),
          "tiles longer than 1024 entries", list(rc.long_tiles_hint.values()))
if os.environ.get("FIT_WALK_STATS", "0") == "1":   # what the backward's walk of one step looks like (vidu4d_surfel_blend_stats)
    from vidu4d_amd import _C
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    _C.count_next_walk(cnt, context=m.raster_context)   # (the model's forwards and backwards run under ITS context)
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
