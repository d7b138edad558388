#!/usr/bin/env python
"""bench.py -- train images/s (fwd+bwd raster) of the MI355X-native surfel rasterizer.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json's headline configuration (configs[2]: Stage-3 gs-bob op, 200 000 surfels,
512x512, SH degree 3, 120 frames), synthetic seeded surfels, inputs resident in HBM.
A *step* is one optimizer step's worth of raster work on a rank: FRAMES_PER_STEP (= 2, the frame
pair `imgs_per_gpu=1` gives the reference, lab4d/engine/trainer.py:439-476) frames, each
forward + backward through the public `GaussianRasterizer` autograd op with upstream gradients on
the colour image and all 8 auxiliary planes.  Frames are sharded one-frame-per-GPU-per-slot across
ranks (rank r renders frames r, r+N, ...: frame-parallel, weak scaling); with N > 1 every step ends
with ONE RCCL all-reduce of the flat canonical-surfel gradient buffer (58 floats per surfel), the
only exchange the path has; it runs beside the next step's kernels on two alternating buffers (the op-level loop has no
optimizer between steps; --exchange serial joins it in its step).  value = N * K * FRAMES_PER_STEP / max-over-ranks time.
As in Stage3Trainer, the frames of a step are queued on separate HIP streams and the rasterizer's
host wait for the pair count is deferred to one check per step (--frame-streams 0: one stream).

`--gpus N` (N > 1) without a torchrun environment spawns the N ranks itself (python -m torch.distributed.run on
127.0.0.1) and relays rank 0's line; it never prints an n_gpus: 1 line for N > 1.  After the contract's timed
region of exactly K steps (-> "value"), `--repeats` further regions of K steps each give "repeats" (median,
p10, p90 of images/s).  "value_per_frame_calls" times the same steps through the reference's own surface (one
GaussianRasterizer.forward per frame).  "fit_step" is the second figure of SURVEY.md 8(d): the full Stage-3 fitting
step (bob warp + raster + losses + backward + clip + densify statistics + Adam) at the same size; "fit_step_geometry"
the same step after step 8000 (normal-consistency regulariser on).  "scaling_modelled" (N = 1 only): MODELLED multi-GPU
speed-ups from the measured step and an all-reduce cost model, assumptions included (fitting steps: also as
Stage3Trainer issues the exchange, the SH rest bands' collective beside the warp's backward).  "settle_steps_before_warmup": untimed steps that precede a short --warmup (up to 60 steps in all,
so that clocks and allocation hints are steady: the driver's --warmup 5 measured 6 % low without them).  "host_enqueue_ms_per_step":
time until the timed region's launches were queued (the host runs into the launch queue's back-pressure: ~ the GPU time).  `--replicas N`: BASELINE
configs[3], N independent sequences with an RCCL barrier at start and end.

Extra objects on the JSON line: "roofline" (dominant kernel: algorithmic bytes per launch /
its average launch duration, measured with HIP events on the launch stream over extra steps of
the same workload (run before the warm-up and the timed region), frames serialised -- under the timed region's
two-stream overlap an event pair does not measure a launch duration) and "cpu_baseline" (the CPU
oracle timed on this box's host cores, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FRAMES_PER_STEP = 2
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); 6.29 TB/s measured copy
GRAD_FLOATS_PER_SURFEL = 3 + 1 + 2 + 4 + 48  # means3D, opacity, scales, rotations, SH


def stage_bytes(stage: str, N: int, R: float, P: int, T: int, K: int) -> float:
    """Algorithmic bytes of one launch of a stage (DESIGN.md 'Measurement')."""
    return {
        "preprocess_fwd": N * (232 + 87),
        "tile_scan": T * 12,
        "emit_keys": N * 20 + R * 12,
        "tile_sort": R * 20,
        
        "blend_fwd": T * 8 + R * 76 + P * 64,
        "bwd_zero": N * 80,
        "blend_bwd": T * 8 + R * 76 + P * 108 + N * 76,
        "preprocess_bwd": N * (347 + 276),
    }[stage]


def baseline_config_label(N: int, W: int, H: int) -> str:
    """Which BASELINE.json configuration a size is (configs[1]: 50 k / 256^2, [2]: 200 k / 512^2 = the headline, [4]: 1 M / 1080p)."""
    return {(50_000, 256, 256): "BASELINE.json configs[1] size (50 k surfels, 256^2)",
            (200_000, 512, 512): "BASELINE.json configs[2] (the headline: 200 k surfels, 512^2)",
            (1_000_000, 1920, 1080): "BASELINE.json configs[4] size (1 M surfels, 1920x1080)"}.get(
                (N, W, H), "a size BASELINE.json does not list")


def higher_msb(n: int) -> int:
    """Bits the reference sorts the tile id on (getHigherMsb, rasterizer_impl.cu:35-50)."""
    msb = step = 16
    while step > 1:
        step //= 2
        msb = msb + step if (n >> msb) else msb - step
    return msb + 1 if (n >> msb) else msb


def total_bytes(N, R, P, T, K):
    """The reference algorithm's bytes per image (K 8-bit radix passes over 24-byte pairs).
    This is the figure the roofline fraction of the whole op is quoted against; our own pipeline moves
    fewer (the tile-binned sort touches each 8-byte pair three times instead of 2K times)."""
    return N * 1046 + R * (172 + 24 * K) + P * 172 + T * 24


def cpu_baseline(scene, n_images: int):
    """The CPU oracle (oracle/surfel_oracle.c, OpenMP over tiles/surfels) on the same workload."""
    from oracle import surfel_oracle as so
    from vidu4d_amd.synthetic import frame_motion, make_upstream_grads
    cores = so.set_threads(min(64, os.cpu_count() or 1))  # the tile loop stops scaling beyond ~64 threads
    dc, do = make_upstream_grads(scene.width, scene.height)
    t_fwd = t_all = 0.0
    for f in range(n_images + 1):
        sc = frame_motion(scene, f, 120)
        t0 = time.perf_counter()
        st = so.forward(sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.viewmatrix, sc.projmatrix, sc.campos,
                        sc.bg, sc.width, sc.height, sc.tanfovx, sc.tanfovy, sc.sh_degree, shs=sc.shs)
        t1 = time.perf_counter()
        so.backward(st, dc, do)
        t2 = time.perf_counter()
        if f > 0:  # first image is the warm-up
            t_fwd += t1 - t0
            t_all += t2 - t0
    return {"value": n_images / t_all, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_images} frames fwd+bwd of the same {scene.num_surfels}-surfel {scene.width}x{scene.height} "
                      f"workload (1 warm-up frame untimed), C oracle with OpenMP",
            "fwd_only_images_per_s": n_images / t_fwd}


def fit_step_rate(dev, n_surfels: int, W: int, H: int, steps: int, start_step: int = 0, seed: int = 0, barrier=None,
                  densify: bool = False, optim_warp: bool = False, fused_warp_trainable: bool = True, captured="auto"):
    """Second figure of SURVEY.md 8(d): images/s of the FULL Stage-3 fitting step (bob LBS warp with frozen,
    randomly initialised warp / camera networks -> rasterize 2 frames -> losses -> backward -> gradient clip ->
    densification statistics -> Adam) on an object-centric synthetic sequence of the same size.
    start_step > 8000: the regularised regime of the second third of a Stage-3 run (lambda_normal on: all 8 planes
    through the blend kernels, depth / normal post-processing + normal-consistency term inside the loss kernels;
    BASELINE.json configs[4] "depth/normal reg on").
    densify: BASELINE.json configs[2] as written -- "densify+prune on": the schedule's densify_and_prune every 100 steps
    from step 500 on (lab4d/engine/trainer.py:549-572, gs/scene/gaussian_model.py:434-448) runs INSIDE the timed region
    (start_step 501 and >= 300 steps: three events), on the device (csrc/optim.hip); the surfel count before / after is
    reported.
    optim_warp: the reference's DEFAULT flag value --gs_optim_warp=True (lab4d/config.py:157): the bone / articulation / camera /
    skinning networks train too (AdamW, trainer.py:592-598; start_step >= optim_warp_neus_iters = 12 000 so that it steps).
    Round 5: the fused warp then still applies -- the networks are evaluated for the step's frames with autograd, the delta-skin
    MLP as library GEMMs, and the skinning kernel's backward reduces d/d (bone dual quaternions, cameras) over the surfels;
    fused_warp_trainable=False times the ~40-kernel torch chain rounds 1-4 fell back to (same step, for the A/B).
    captured (round 6): "auto" = the trainer's default -- plain steps replayed from ONE captured hipGraph
    (lab4d/captured_step.py) when the networks train, the eager loop when they are frozen (GPU-bound there: the graph's
    per-node cost makes it slower); True / False force it on / off for the A/B lines."""
    import numpy as np
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    frames = 120
    m = DeformableSurfels(dict(fg_motion="gs-bob", captured_step=captured) | ({} if densify else dict(densify_until_iter=0)) |
                          (dict(fused_warp_trainable=fused_warp_trainable) if optim_warp else {}), num_frames=frames, device=dev)
    d = rng.normal(size=(n_surfels, 3)).astype(np.float32)
    pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.2, 1.0, size=(n_surfels, 1)).astype(np.float32) ** (1 / 3)
    m.init_from_points(pts.astype(np.float32), rng.uniform(size=(n_surfels, 3)).astype(np.float32))
    tr = Stage3Trainer(m, (m.opts | dict(gs_optim_warp=True)) if optim_warp else None)
    if optim_warp:
        assert m.warp_networks_train() and m.fused_warp_ok() == bool(fused_warp_trainable)
    if start_step:
        m.active_sh_degree = m.max_sh_degree   # (reached at step 3000)
        tr.current_steps = start_step
    batches = [synthetic_batch(m, [(2 * i) % frames, (2 * i + 1) % frames], H, W, seed=i) for i in range(8)]
    warm = 20   # (capacity / segment-depth hints and the allocator settle over the first ~10 steps: with 6 warm-up steps
    #             and 30 timed ones round 3's figure sat 18 % below what the same loop gives over 100 steps, tools/fit_profile.py)
    for i in range(warm):
        tr.train_step(batches[i % 8])
    if densify:
        tr.current_steps = start_step   # (the warm-up steps do not count towards the cadence)
    torch.cuda.synchronize(dev)
    if barrier is not None:
        barrier()
    n_before, events = int(m._xyz.shape[0]), []
    t0 = time.perf_counter()
    for i in range(steps):
        n0 = int(m._xyz.shape[0])
        tr.train_step(batches[i % 8])
        if densify and int(m._xyz.shape[0]) != n0:
            events.append((tr.current_steps - 1, n0, int(m._xyz.shape[0])))
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    regime = ("losses as the reference's --rgb_loss_only run before step 8000 (colour + silhouette, lambda_dist = 0), so the "
              "blend kernels run their colour + alpha-plane instances (aux_planes); the op-level `value` above drives all 8 planes"
              if start_step <= 8000 else
              f"steps {start_step}.. of the schedule: normal-consistency regulariser on (lambda_normal = 0.05, model.py:817-842) "
              "with the upstream defaults lambda_dist = 0 and depth_ratio = 0, so only planes 0-4 (depth, alpha, normal) are read: "
              "the blend kernels run their colour + planes-0-4 instances (aux_planes = AUX_GEOM; no median sample, no distortion "
              "moments, no transmittance pre-pass), depth / normal post-processing inside the loss kernels")
    out = {"images_per_s": 2.0 / dt, "ms_per_step": 1e3 * dt, "frames_per_step": 2, "steps": steps, "warmup_steps": warm,
           "seconds": dt * steps, "captured_step": dict(tr.captured_stats, enabled=bool(tr.captured_step)),
           "config": f"{n_surfels} surfels in a unit ball 3 units from the camera, {W}x{H}, 25 bones, 120 frames, "
                     + ("warp / camera / skinning networks TRAIN (--gs_optim_warp=True, the reference's default; AdamW steps on them), " +
                        ("fused warp with parameter gradients from the skinning kernel" if fused_warp_trainable else
                         "un-fused torch warp chain (rounds 1-4's fallback)") if optim_warp else
                        "warp / camera networks frozen (--gs_optim_warp=False)") +
                     ", densify " + ("+ prune ON; " if densify else "off; ") + regime}
    if densify:
        out["surfels_before"], out["surfels_after"] = n_before, int(m._xyz.shape[0])
        out["densification_events"] = [{"step": s, "surfels": [a, b]} for s, a, b in events]
        out["steps_of_the_schedule"] = [start_step, start_step + steps]
    return out


# xGMI: 7 links per GPU, ~153 GB/s each (task statement / MI355X guide); one kernel-boundary-sized latency per collective phase
XGMI_LINKS, XGMI_LINK_GBPS, COLLECTIVE_LATENCY_MS = 7, 153.0, 0.02


def exchange_model_ms(payload_bytes: float, n: int) -> dict:
    """MODELLED time of the step's one all-reduce of `payload_bytes` over n GPUs of one node (no multi-GPU node was
    available to measure it).  Three algorithms bracket what RCCL does on a fully connected xGMI node:
      ring_one_link   a single ring, each hop over ONE link: 2 (n-1)/n S / 153 GB/s            (pessimistic)
      ring_all_links  RCCL's multi-ring schedule over all 7 links at 70 % of their sum            (typical, large messages)
      direct_rs_ag    reduce-scatter + all-gather with every peer at once, S/n per link per phase (what xGMI's
                      point-to-point topology allows)"""
    if n <= 1:
        return {"ring_one_link": 0.0, "ring_all_links": 0.0, "direct_rs_ag": 0.0}
    s = float(payload_bytes)
    lat = COLLECTIVE_LATENCY_MS
    links = min(XGMI_LINKS, n - 1)              # links of one GPU that lead to a peer inside the group
    eff = 0.7 if links > 1 else 1.0
    wire = 2.0 * (n - 1) / n * s                # bytes every GPU sends (and receives) in a ring all-reduce
    return {"ring_one_link": wire / (XGMI_LINK_GBPS * 1e9) * 1e3 + 2 * lat,
            "ring_all_links": wire / (eff * links * XGMI_LINK_GBPS * 1e9) * 1e3 + 2 * lat,
            "direct_rs_ag": 2.0 * (s / n) / (XGMI_LINK_GBPS * 1e9) * 1e3 + 2 * lat}


def scaling_model(ms_per_step_1gpu: float, payload_bytes: float, overlapped: bool, contention: float = 1.0) -> dict:
    """MODELLED weak scaling (images/s at N GPUs / images/s at 1 GPU) from the MEASURED 1-GPU step and the exchange
    model above.  serial: the exchange sits between backward and optimizer (the fitting loop: Adam needs the reduced
    gradient); overlapped: it runs beside the next step's kernels (bench.py's op-level loop, double-buffered gradient
    buffers) and costs only what exceeds a step."""
    out = {}
    for n in (2, 4, 8):
        ex = exchange_model_ms(payload_bytes, n)
        # (overlapped: the step runs `contention` times slower while a collective shares the GPU with it)
        out[str(n)] = {k: round(n * ms_per_step_1gpu / (max(ms_per_step_1gpu * contention, v) if overlapped
                                                        else ms_per_step_1gpu + v), 2) for k, v in ex.items()}
        out[str(n)]["exchange_ms"] = {k: round(v, 3) for k, v in ex.items()}
    return out


# what of the fitting step follows the rasterizer's backward on the compute stream before the optimizer needs the
# gradients: the warp's backward (lbs_skin 44 us + skin_field 64 us + its torch glue) -- rocprofv3, profiles/r03_fit_step_*
WARP_BACKWARD_MS = 0.15


def measured_contention() -> dict:
    """What a collective's kernel costs the kernels it shares the GPU with, MEASURED on one GPU (tools/contention_probe.py ->
    profiles/r04_contention.json): the step's 46.4 MB copied device-to-device by 16 / 32 / 64 workgroups on a side stream
    while the headline backward runs.  {"factor": slowdown of the overlapped compute, ...}; factor 1.0 if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r04_contention.json")
    try:
        d = json.load(open(path))
        f = max(v for k, v in d["backward_slowdown"].items() if not k.startswith("0"))
        return {"factor": round(float(f), 4), "source": "profiles/r04_contention.json (tools/contention_probe.py)",
                "what": "backward of the headline step (blend_bwd + preprocess_bwd) with a 46.4 MB device-to-device copy "
                        "looping on 16-64 workgroups of a side stream / without: the worst of the three tenant sizes"}
    except Exception:
        return {"factor": 1.0, "source": "not measured (profiles/r04_contention.json absent)"}


def fit_scaling_model(ms_per_step_1gpu: float, n_surfels: int, contention: float = 1.0) -> dict:
    """MODELLED weak scaling of the fitting step as Stage3Trainer exchanges: the SH rest bands (45 of the 58 floats per
    surfel) go on the wire behind the rasterizer's backward and ride beside the warp's backward (WARP_BACKWARD_MS of
    compute that does not touch them, slowed by the MEASURED co-tenant factor `contention`); the 13 small floats per
    surfel (+ background) follow behind the whole backward.
    Serial cost = what of the first collective outlasts the warp's backward + the second collective."""
    out = {}
    rest, small = n_surfels * 45 * 4, n_surfels * 13 * 4 + 12
    for n in (2, 4, 8):
        e_rest, e_small = exchange_model_ms(rest, n), exchange_model_ms(small, n)
        serial = {k: max(0.0, e_rest[k] - WARP_BACKWARD_MS) + e_small[k] + (contention - 1.0) * WARP_BACKWARD_MS for k in e_rest}
        out[str(n)] = {k: round(n * ms_per_step_1gpu / (ms_per_step_1gpu + v), 2) for k, v in serial.items()}
        out[str(n)]["serial_exchange_ms"] = {k: round(v, 3) for k, v in serial.items()}
    return out


def torch_cpu_baseline(scene, n_images: int, threads: int):
    """The pure-PyTorch CPU render BASELINE.json's north_star asks to be timed beside the GPU number
    (oracle/torch_render.py: vectorised forward, autograd backward)."""
    from oracle import torch_render as tr
    from vidu4d_amd.synthetic import make_upstream_grads
    torch.set_num_threads(threads)
    dc, do = make_upstream_grads(scene.width, scene.height)
    t_all = 0.0
    for f in range(n_images):
        ins = [x.clone().requires_grad_(True) for x in (scene.means3D, scene.opacities, scene.scales,
                                                         scene.rotations, scene.shs)]
        t0 = time.perf_counter()
        color, radii, others, state = tr.rasterize(ins[0], ins[1], ins[2], ins[3], scene.viewmatrix, scene.campos,
                                                   scene.bg, scene.width, scene.height, scene.tanfovx, scene.tanfovy,
                                                   scene.sh_degree, shs=ins[4])
        ((color * dc).sum() + (others * do).sum()).backward()
        t_all += time.perf_counter() - t0
    return {"value": n_images / t_all, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{n_images} frame(s) fwd+bwd, pure-PyTorch vectorised render + autograd, "
                      f"{scene.num_surfels} surfels {scene.width}x{scene.height}"}


def torch_cpu_baseline_bounded(args, limit_s: float = 150.0):
    """Runs the PyTorch CPU baseline in a child process so that a slow host cannot hold up the bench:
    past `limit_s` the child is killed and the field reports the bound instead of a rate."""
    import subprocess
    threads = min(32, os.cpu_count() or 1)  # beyond ~32 threads the many small tensor ops only contend
    cmd = [sys.executable, os.path.abspath(__file__), "--_torch_cpu_child", str(args.torch_cpu_images), "--surfels",
           str(args.surfels), "--res", str(args.res), "--height", str(args.height), "--_threads", str(threads)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if line else {"value": None, "error": (r.stderr or "no output")[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"did not finish {args.torch_cpu_images} frame(s) within {limit_s:.0f} s (< "
                          f"{args.torch_cpu_images / limit_s:.4f} images/s)"}


def host_cost_probe(limit_s: float = 120.0) -> dict:
    """What the HOST needs per step, measured where it can be seen: the same bench loop on a scene whose kernels take next to
    nothing (2 000 surfels, 64^2), so ms_per_step IS the Python + launch cost of a step -- through the stacked surface, the
    reference's per-frame surface on two streams, and on one.  (`host_enqueue_ms_per_step` of the headline run equals its GPU
    time whenever the GPU is the slower side: the host then waits in the launch queue.)"""
    import subprocess
    out = {}
    cmd = [sys.executable, os.path.abspath(__file__), "--surfels", "2000", "--res", "64", "--steps", "1000", "--warmup", "100",
           "--cpu-images", "0", "--torch-cpu-images", "0", "--fit-steps", "0", "--repeats", "0", "--per-frame-surface", "1",
           "--no-stage-timers", "--host-probe", "0"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if line:
            d = json.loads(line[-1])
            pf = d.get("value_per_frame_calls", {})
            out = {"stacked_ms_per_step": round(d["ms_per_step"], 4),
                   "per_frame_two_streams_ms_per_step": round(pf.get("ms_per_step", float("nan")), 4),
                   "per_frame_one_stream_ms_per_step": round(pf.get("single_stream", {}).get("ms_per_step", float("nan")), 4)}
    except subprocess.TimeoutExpired:
        out = {"error": f"did not finish within {limit_s:.0f} s"}
    out["what"] = "bench.py's own step loop at 2 000 surfels / 64^2 (kernels ~ nothing): Python + launch cost of a 2-frame step"
    return out


def replicas_main(args, world, rank, local_rank):
    """BASELINE.json configs[3]: `world` independent sequences, one process per GPU, nothing exchanged -- an RCCL
    barrier before the timed loop and one after it (the pattern of lab4d/utils/gpu_utils.py:6-128 /
    scripts/run_rendering_parallel.py:47-68: a pool of per-GPU workers).  Each rank fits ITS OWN sequence (own seed);
    value = world * steps * 2 frames / the slowest rank's time."""
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29513")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    steps = max(1, args.fit_steps)
    r = fit_step_rate(dev, args.surfels, args.res, args.height or args.res, steps, seed=rank, barrier=dist.barrier)
    dist.barrier()
    tt = torch.tensor([r["seconds"]], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        secs = float(tt.item())
        out = {"metric": f"train images/sec (full Stage-3 fitting step), {world} independent sequence(s) one per GPU",
               "value": world * steps * 2 / secs, "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": 6,
               "ms_per_step": 1e3 * secs / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[3]: independent sequences, 1 per GPU, RCCL barrier at start / end only; "
                                      + r["config"], "parallelism": f"replicas x{world}"}}
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--surfels", type=int, default=200_000)
    ap.add_argument("--res", type=int, default=512, help="image width (and height unless --height is given)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--scene", choices=["uniform", "object"], default="uniform",
                    help="uniform: BASELINE.md scene (surfels spread over the frustum); object: the same surfels "
                         "inside a ball covering about a third of the image (uneven tile lists)")
    ap.add_argument("--object-radius", type=float, default=1.0)
    ap.add_argument("--opacity", choices=["random", "init"], default="random",
                    help="random: sigmoid(N(0, 2^2)) (BASELINE.md scene); init: every surfel 0.1, the Stage-3 initialisation "
                         "(gs/scene/gaussian_model.py:143) -- nothing saturates, every list is walked to its end (SURVEY.md 8d)")
    ap.add_argument("--cpu-images", type=int, default=6, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--torch-cpu-images", type=int, default=1,
                    help="frames timed on the pure-PyTorch CPU render (0 = skip; ~10-20 s each at 200k/512^2)")
    ap.add_argument("--no-stage-timers", action="store_true")
    ap.add_argument("--frame-streams", type=int, default=1, help="queue the frames of a step on separate HIP streams")
    ap.add_argument("--stacked", type=int, default=1, help="1 (default): the frames of a step through ONE launch set "
                                                            "(dsr.rasterize_frames, frame || tile keys); 0: one call per "
                                                            "frame, on separate HIP streams with --frame-streams 1")
    ap.add_argument("--repeats", type=int, default=5, help="further timed regions of --steps steps (median / p10 / p90)")
    ap.add_argument("--fit-steps", type=int, default=60, help="steps of the full Stage-3 fitting loop timed for "
                                                              "\"fit_step\" and \"fit_step_geometry\" (0 = skip)")
    ap.add_argument("--fit-densify-steps", type=int, default=320, help="steps of the fitting loop timed with densify + prune "
                    "ON from step 501 of the schedule (three densification events; BASELINE.json configs[2] verbatim; 0 = skip)")
    ap.add_argument("--host-probe", type=int, default=1, help="also measure the host's own cost of a step (a child run on a "
                    "scene whose kernels take next to nothing) -> \"host_cost\"")
    ap.add_argument("--fit-optim-warp", type=int, default=1, help="also time the fitting step with TRAINING warp / camera networks "
                    "(--gs_optim_warp=True, the reference's default) -> \"fit_step_optim_warp\" and its un-fused A/B")
    ap.add_argument("--exchange", choices=["overlapped", "serial"], default="overlapped",
                    help="N > 1: overlapped = the step's all-reduce runs beside the next step's kernels (two gradient "
                         "buffers, joined before its buffer is reused: the op-level loop has no optimizer between steps); "
                         "serial = joined at the end of its step")
    ap.add_argument("--per-frame-surface", type=int, default=1,
                    help="also time the reference's per-frame GaussianRasterizer.forward surface (one call per frame) "
                         "-> \"value_per_frame_calls\"")
    ap.add_argument("--replicas", type=int, default=0,
                    help="BASELINE.json configs[3]: N independent Stage-3 sequences, one per GPU, RCCL barrier at start "
                         "and end only; prints the aggregate images/s of the full fitting step")
    ap.add_argument("--_torch_cpu_child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--_threads", type=int, default=8, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args._torch_cpu_child:
        from vidu4d_amd.synthetic import make_scene
        print(json.dumps(torch_cpu_baseline(make_scene(args.surfels, args.res, args.height or None, seed=1234),
                                            args._torch_cpu_child,
                                            args._threads)), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.replicas:
        args.gpus = args.replicas
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if args.replicas and (args.replicas == 1 or "WORLD_SIZE" in os.environ):
        return replicas_main(args, world, rank, local_rank)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start the ranks ourselves (one process per GPU over RCCL) and relay rank 0's line
        import subprocess
        # VIDU4D_BENCH_BACKEND=gloo (CI on one GPU): the ranks may share a device -- gloo stages device tensors through the
        # host, RCCL refuses two ranks on one GPU.  Not a performance configuration: it runs the N > 1 code path.
        if torch.cuda.device_count() < args.gpus and os.environ.get("VIDU4D_BENCH_BACKEND", "nccl") != "gloo":
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
            raise SystemExit(f"spawning {args.gpus} ranks failed (rc {r.returncode})")
        print(lines[-1], flush=True)
        return
    backend = os.environ.get("VIDU4D_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # VIDU4D_BENCH_FORCE_DIST=1 exercises the collective path with a 1-rank RCCL group (CI on one GPU)
    use_dist = world > 1 or os.environ.get("VIDU4D_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if use_dist and backend != "gloo" and torch.cuda.current_device() != local_rank:
        raise SystemExit(f"rank {rank}: HIP device {torch.cuda.current_device()} is not LOCAL_RANK {local_rank}")

    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _lib
    from vidu4d_amd.synthetic import frame_motion, make_object_scene, make_scene, make_upstream_grads

    N, W = args.surfels, args.res
    if args.scene == "object":
        scene_cpu = make_object_scene(N, W, args.height or None, radius=args.object_radius, seed=1234, opacity_mode=args.opacity)
    else:
        scene_cpu = make_scene(N, W, args.height or None, seed=1234, opacity_mode=args.opacity)
    scene = scene_cpu.to(dev)
    H = scene.height
    dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
    # this rank's frames, resident in HBM before the timed region
    my_frames = list(range(rank, args.frames, world))
    frames = [frame_motion(scene, f, args.frames) for f in my_frames]
    means = [f.means3D for f in frames]
    rots = [f.rotations for f in frames]
    rs = dsr.GaussianRasterizationSettings(H, W, scene.tanfovx, scene.tanfovy, scene.bg, 1.0, scene.viewmatrix,
                                           scene.projmatrix, scene.sh_degree, scene.campos, False, False)
    rast = dsr.GaussianRasterizer(rs)
    opac = scene.opacities.clone().requires_grad_(True)
    scales = scene.scales.clone().requires_grad_(True)
    shs = scene.shs.clone().requires_grad_(True)
    # the step's gradients are produced into one of TWO persistent flat buffers, alternately, so that the all-reduce of
    # step i can still be on the wire while step i + 1 fills the other one (--exchange overlapped)
    flats = [torch.empty(N * GRAD_FLOATS_PER_SURFEL, device=dev) for _ in range(2)] if use_dist else None
    pending = [None, None]
    # the first N > 1 launch validates itself before anything is timed (vidu4d_amd/lab4d/dist_check.py): every rank seen,
    # device == LOCAL_RANK, the payload's standalone all-reduce time, which frames each rank renders
    rccl_check = None
    if use_dist:
        from vidu4d_amd.lab4d.dist_check import collective_self_check
        rccl_check = collective_self_check(dist, dev, local_rank, flats[0], my_frames, backend=backend)
    use_distributed_exchange = use_dist
    opac_f, scales_f, shs_f = opac, scales, shs
    counter = {"slot": 0, "R": 0.0, "n": 0, "buf": 0}

    # The frames of a step are independent until their gradients are summed: like Stage3Trainer, each one
    # is queued on its own HIP stream, so the tail of one frame's blend kernels overlaps the next frame's
    # projection / binning instead of leaving CUs idle (--frame-streams 0: one stream).
    mode = {"streams": bool(args.frame_streams), "stacked": bool(args.stacked)}
    side = [torch.cuda.Stream(device=dev) for _ in range(FRAMES_PER_STEP)]

    def one_frame(i):
        m = means[i].detach().requires_grad_(True)
        r = rots[i].detach().requires_grad_(True)
        m2d = torch.zeros_like(m, requires_grad=True)
        color, radii, allmap = rast(means3D=m, means2D=m2d, opacities=opac_f, shs=shs_f, scales=scales_f, rotations=r)
        torch.autograd.backward([color, allmap], [dc, do])
        return m.grad, r.grad

    from vidu4d_amd import _C as native

    def step():
        # (like Stage3Trainer) the rasterizer's host wait for the pair count is deferred to one check per
        # step; a frame that outgrew its binning buffer would have rendered only its background, so the
        # step is then repeated -- it never happens after the warm-up, and it is inside the timed region
        with native.deferred_capacity_check():
            grads = step_once()
        if not native.check_deferred():
            grads = step_once()
        # the path's only exchange: canonical-surfel gradients, once per optimizer step.  (After the capacity
        # check, so that a repeated step on one rank cannot add a collective the other ranks do not make.)
        if use_distributed_exchange:
            # (the shared-parameter gradients already live in the flat buffer: their .grad tensors are views of it)
            b = counter["buf"]
            pending[b] = dist.all_reduce(flats[b], async_op=True)   # RCCL's stream waits for the kernels queued so far
            if args.exchange == "serial":
                pending[b].wait()
                pending[b] = None
            counter["buf"] = b ^ 1

    if use_dist:
        # one persistent flat exchange buffer; the shared parameters' .grad are views of it (autograd accumulates
        # into an existing .grad in place), the per-frame means / rotations gradients are summed straight into it
        offs = {}
        o = 0
        for name, n in (("means", N * 3), ("opac", N), ("scales", N * 2), ("rot", N * 4), ("shs", N * 48)):
            offs[name] = (o, o + n)
            o += n
        flat_views = [{k: f[a:b] for k, (a, b) in offs.items()} for f in flats]

    def claim_flat():
        """This step's gradient buffer, zeroed, with the shared parameters' .grad bound to it; joins the collective that
        last used it (two steps ago)."""
        b = counter["buf"]
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        fv = flat_views[b]
        # opacity / scale / SH gradients are written by the rasterizer's backward straight into the buffer (one stacked
        # call produces each of them exactly once: _C.gradient_buffers) and adopted by autograd as .grad; the per-frame
        # surface has two calls per step and accumulates the second into them, so it starts from zeros
        for t in (opac, scales, shs):
            t.grad = None
        if not mode["stacked"]:
            flats[b].zero_()
            opac.grad = fv["opac"].view_as(opac)
            scales.grad = fv["scales"].view_as(scales)
            shs.grad = fv["shs"].view_as(shs)
        return fv

    rs_frames = [rs] * FRAMES_PER_STEP
    dc_st, do_st = torch.stack([dc] * FRAMES_PER_STEP, 1).contiguous(), torch.stack([do] * FRAMES_PER_STEP, 1).contiguous()

    stacked_inputs = {}
    screen_dummy = [torch.zeros(FRAMES_PER_STEP, N, 3, device=dev)]

    def stacked_frames():
        """The step's frames as one stacked call (frame || tile keys)."""
        ids = tuple((counter["slot"] + k) % len(frames) for k in range(FRAMES_PER_STEP))
        counter["slot"] += FRAMES_PER_STEP
        # the step's frames as (F, N, .) tensors: stacked once per frame pair (they are inputs: resident in HBM before the
        # timed region, like the per-frame tensors), not copied every step
        if ids not in stacked_inputs:
            stacked_inputs[ids] = (torch.stack([means[i] for i in ids]), torch.stack([rots[i] for i in ids]))
        m, r = (t.detach().requires_grad_(True) for t in stacked_inputs[ids])
        # means2D is a dummy whose .grad receives the densification statistic; its values are never read (upstream fills a
        # fresh zeros tensor per frame, gaussian_renderer/__init__.py:29): one persistent buffer, as Stage3Trainer does
        m2d = screen_dummy[0].detach().requires_grad_(True)
        color, radii, allmap = dsr.rasterize_frames(m, m2d, shs_f, opac_f, scales_f, r, rs_frames)
        torch.autograd.backward([color, allmap], [dc_st, do_st])
        return m.grad, r.grad

    def step_once():
        if use_dist:
            flat_view = claim_flat()
        else:
            for t in (opac, scales, shs):
                t.grad = None
        if mode["stacked"]:
            if use_dist:
                with native.gradient_buffers(dL_dopacity=flat_view["opac"].view_as(opac), dL_dscales=flat_view["scales"].view_as(scales),
                                             dL_dsh=flat_view["shs"].view_as(shs)):
                    gm, gr = stacked_frames()
            else:
                gm, gr = stacked_frames()
            if use_dist:
                # (the exchanged payload carries the canonical centres' / orientations' gradients: the frames' sums)
                torch.sum(gm, 0, out=flat_view["means"].view(N, 3))
                torch.sum(gr, 0, out=flat_view["rot"].view(N, 4))
                return None
            return gm, gr   # (per-frame gradients, as the op returns them: in the fitting loop they enter the warp's backward)
        use_streams = mode["streams"]
        main = torch.cuda.current_stream(dev)
        ready = main.record_event() if use_streams else None
        per_frame = []
        for k in range(FRAMES_PER_STEP):
            i = counter["slot"] % len(frames)
            counter["slot"] += 1
            if use_streams:
                side[k].wait_event(ready)
                with torch.cuda.stream(side[k]):
                    per_frame.append(one_frame(i))
            else:
                per_frame.append(one_frame(i))
        if use_streams:
            for st in side:
                main.wait_stream(st)
        if use_dist:
            torch.add(per_frame[0][0], per_frame[1][0], out=flat_view["means"].view(N, 3))
            torch.add(per_frame[0][1], per_frame[1][1], out=flat_view["rot"].view(N, 4))
            return None
        return sum(g[0] for g in per_frame), sum(g[1] for g in per_frame)

    def sync():
        for b in range(2):   # (an all-reduce still in flight belongs to the region that issued it)
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- Per-kernel launch durations for the roofline: HIP events on the launch stream around every stage
    # (vidu4d_surfel_profile_*), over steps of the same workload run BEFORE the warm-up and the timed region, with
    # the frames of a step queued one after the other.  In the timed region two frames overlap on two streams, and an event
    # pair around a kernel then measures that kernel sharing the GPU with another frame's kernels -- not
    # a launch duration (rocprofv3's kernel trace, which serialises, would not agree with it either).
    if not args.no_stage_timers:
        mode["streams"] = False
        for _ in range(3):  # (capacity hints settle, allocator pools fill)
            step()
        sync()
        _lib.profile_read(reset=True)
        _lib.profile_enable(True)
        for _ in range(min(args.steps, 20)):
            step()
        sync()
        _lib.profile_enable(False)
        mode["streams"] = bool(args.frame_streams)
    # (the cyclic collector is kept out of the timed regions: a generation-2 pass over torch's Python objects takes
    # milliseconds, as long as several steps.  Collected BEFORE the warm-up: between the warm-up and the timed region
    # there is then nothing but the barrier + synchronize the contract prescribes -- with the collection in between the
    # GPU sat idle for its duration and the first timed region measured 2.5 % below the repeats that follow it.)
    import gc
    gc.collect()
    gc.disable()
    # settle: a short --warmup (the driver's 5) ends before the clocks and the allocator / capacity hints are steady -- a
    # 20-step region right behind it measured 6 % below the default run's.  Untimed steps up to 60 in all come first; the
    # contract's W warm-up steps and the barrier + synchronize follow as prescribed.
    settle = max(0, 60 - args.warmup)
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    enqueued = time.perf_counter() - t0   # (host side: all launches of the region queued; the GPU is still working)
    sync()
    elapsed = time.perf_counter() - t0
    rep_rates = []
    for _ in range(max(0, args.repeats)):
        sync()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        rep_rates.append(world * args.steps * FRAMES_PER_STEP / (time.perf_counter() - r0))
    # ---- the reference's own surface: one GaussianRasterizer.forward call per frame (here: on separate HIP streams),
    # timed the same way -- `value` above runs the frames of a step through ONE stacked launch set (an extension)
    per_frame = None
    if args.stacked and args.per_frame_surface:
        mode["stacked"] = False
        for _ in range(min(args.warmup, 10)):
            step()
        sync()
        p0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        pf_elapsed = time.perf_counter() - p0
        pf_rates = []
        for _ in range(max(0, args.repeats)):
            sync()
            r0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            pf_rates.append(world * args.steps * FRAMES_PER_STEP / (time.perf_counter() - r0))
        # ... and on ONE stream, frame after frame: what a caller that does not touch HIP streams gets -- the reference's
        # own loop (deformable_gaussian.py:1175-1228) unmodified
        mode["streams"] = False
        for _ in range(min(args.warmup, 10)):
            step()
        sync()
        p1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        pf1_elapsed = time.perf_counter() - p1
        mode["streams"] = bool(args.frame_streams)
        mode["stacked"] = True
        per_frame = (pf_elapsed, pf_rates, pf1_elapsed)
    gc.enable()
    if use_dist:
        tt = torch.tensor([elapsed, per_frame[0] if per_frame else 0.0, per_frame[2] if per_frame else 0.0], device=dev,
                          dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        if per_frame:
            per_frame = (float(tt[1].item()), per_frame[1], float(tt[2].item()))
        # every rank's own host time of the timed region (N Python processes share the node's cores: the host side is
        # what a frame-parallel launch multiplies, and what no single-GPU run can show)
        hh = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(hh, torch.tensor([1e3 * enqueued / args.steps], dtype=torch.float64, device=dev))
        host_ms_of_each_rank = [round(float(h.item()), 4) for h in hh]
    else:
        host_ms_of_each_rank = None

    # what the dominant kernel's tile walk looks like on this workload (a counting pass behind one extra step, outside
    # every timed region): lane utilisation = contributing (pixel, surfel) pairs / (64 lanes x pair evaluations)
    walk = None
    lib = _lib.load()
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    if rank == 0:
        from vidu4d_amd import _C as _Cmod
        _Cmod.count_next_walk(cnt)   # (per call: the step's first backward carries the counters)
    step()          # (every rank: the step carries the collective)
    sync()
    if rank == 0:
        c = [int(x) for x in cnt.tolist()]
        if c[1]:
            # (a trip = one pass of a wave through its loop body: one list entry per 32-lane half, the "half walk")
            walk = {"list_entries_staged": c[0], "wave_trips": c[1], "trips_with_a_contributor": c[2],
                    "contributing_pairs": c[3], "lane_utilisation": c[3] / (64.0 * c[1]),
                    "rows_touched_per_contributing_trip": c[4] / max(1, c[2]),
                    "contributing_lanes_histogram_le_4_8_16_32_64": c[5:10],
                    "trips_whose_footprints_miss_their_blocks": c[10], "per": "step (its frames, one or more launches)"}

    images = world * args.steps * FRAMES_PER_STEP
    value = images / elapsed
    out = {
        "metric": ("train images/sec (fwd+bwd raster) @200k surfels, 512^2" if (N, W) == (200_000, 512)
                   else f"train images/sec (fwd+bwd raster) @{N} surfels, {W}x{H}") +
                  (f" [object-centric scene, radius {args.object_radius}]" if args.scene == "object" else ""),
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "host_enqueue_ms_per_step": 1e3 * enqueued / args.steps, "host_enqueue_ms_per_step_of_each_rank": host_ms_of_each_rank, "settle_steps_before_warmup": settle, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{baseline_config_label(N, W, H)} op-level: Stage-3 gs-bob rasterizer fwd+bwd, {N} surfels, "
                               f"{W}x{H}, SH degree 3, {args.frames} frames sharded one-frame-per-GPU, "
                               f"{FRAMES_PER_STEP} frames per step per GPU" +
                               (", opacity 0.1 everywhere (Stage-3 initialisation, gaussian_model.py:143)" if args.opacity == "init" else ""),
                   "scene": args.scene, "opacity": args.opacity, "surfels": N, "width": W, "height": H, "frames": args.frames, "frames_per_step": FRAMES_PER_STEP,
                   "parallelism": f"frame-parallel x{world}" + ((" + RCCL" if backend != "gloo" else " + gloo") +
                                                                " all-reduce of surfel grads" if world > 1 else ""),
                   "frames_of_rank0": my_frames[:4] + ["..."],
                   "frames_of_a_step": "one stacked launch set" if args.stacked else
                                       ("one call per frame, separate HIP streams" if args.frame_streams else "one call per frame"),
                   # (the whole-tile forward's paired workgroups, VIDU4D_SCHED_PAIR: tiles longer than K / 4 x the launch's mean
                   # list get two workgroups; this scene's lists are alike, so the op-level launch pairs none -- the fitting
                   # steps' dense frames do)
                   "sched_pair_k": int(native.PAIR_K)},
    }
    if rccl_check is not None:
        out["rccl"] = rccl_check
    if world > 1 or use_dist:
        out["config"]["exchange"] = ("all-reduce of the step's flat gradient buffer joined before its buffer is reused, two steps "
                                     "later (the op-level loop has no optimizer between steps)" if args.exchange == "overlapped"
                                     else "all-reduce joined at the end of its step")

    def summary(rates):
        import statistics
        q = sorted(rates)
        pick = lambda f: q[min(len(q) - 1, max(0, int(round(f * (len(q) - 1)))))]  # noqa: E731
        return {"n": len(q), "unit": "images/s", "median": statistics.median(q), "p10": pick(0.1), "p90": pick(0.9),
                "min": q[0], "max": q[-1], "note": "this rank's clock; each region = --steps steps"}

    if rep_rates:
        out["repeats"] = summary(rep_rates)
    if per_frame:
        out["value_per_frame_calls"] = {
            "value": images / per_frame[0], "unit": "images/s", "ms_per_step": 1e3 * per_frame[0] / args.steps,
            "surface": "GaussianRasterizer.forward once per frame (the reference's call pattern, "
                       "deformable_gaussian.py:1175-1228), frames of a step on separate HIP streams; same steps / barriers "
                       "as `value`", "steps": args.steps}
        if per_frame[1]:
            out["value_per_frame_calls"]["repeats"] = summary(per_frame[1])
        out["value_per_frame_calls"]["single_stream"] = {
            "value": images / per_frame[2], "unit": "images/s", "ms_per_step": 1e3 * per_frame[2] / args.steps,
            "surface": "the same calls on ONE HIP stream, frame after frame (an unmodified caller of the reference's loop)"}
    if rank == 0:
        # ---- roofline of the dominant kernel (live HIP-event stage timers, see above)
        from vidu4d_amd import _C
        # num_rendered of this rank's frames (read once, outside the timed region)
        Rs, longest = [], 0
        with torch.no_grad():
            for i in range(min(len(frames), 4)):
                o = _C.rasterize_gaussians(scene.bg, means[i], torch.empty(0, device=dev), scene.opacities,
                                           scene.scales, rots[i], 1.0, torch.empty(0, device=dev), scene.viewmatrix,
                                           scene.projmatrix, scene.tanfovx, scene.tanfovy, H, W, scene.shs, 3,
                                           scene.campos, False, False)
                Rs.append(o[0])
                longest = max(longest, int(o[4][:16].view(torch.int32)[2]))  # Header::max_tile_len
        R = sum(Rs) / len(Rs)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        K = (32 + higher_msb(T) + 7) // 8  # 8-bit radix passes over the (tile | depth) key
        out["config"]["num_rendered_mean"] = R
        out["config"]["longest_tile_list"] = longest
        out["config"]["algorithmic_bytes_per_image"] = total_bytes(N, R, W * H, T, K)
        out["algorithmic_GBps_whole_op"] = total_bytes(N, R, W * H, T, K) * (images / world) / elapsed / 1e9
        if not args.no_stage_timers:
            prof = _lib.profile_read(reset=True)
            stages = {k: {"ms_total": ms, "launches": n, "ms_avg": (ms / n if n else None)} for k, (ms, n) in prof.items()}
            out["stage_ms_avg"] = {k: (round(v["ms_avg"], 4) if v["ms_avg"] is not None else None) for k, v in stages.items()}
            # (a stacked launch covers all frames of a step: divide by this for per-frame kernel time)
            out["stage_frames_per_launch"] = FRAMES_PER_STEP if args.stacked else 1
            dom = max((k for k in stages if stages[k]["launches"]), key=lambda k: stages[k]["ms_total"])
            avg_s = stages[dom]["ms_avg"] * 1e-3
            # units one launch processes: the frames of a step when they go through one stacked launch set
            units = FRAMES_PER_STEP if args.stacked else 1
            launch_bytes = units * stage_bytes(dom, N, R, W * H, T, K)
            ach = launch_bytes / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            # `traffic` and the SQ / TCC part of `limiter` are NOT measured by this run (counters need rocprofv3 passes of
            # their own): they are REPLAYED from the committed summary of such passes, and the line says from which file,
            # tree and day -- `achieved`, `avg_launch_ms` and `limiter.tile_walk` are this run's own
            replayed_from = None
            if os.path.exists(tpath):
                try:
                    pj = json.load(open(tpath))
                    traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
                    meta = pj.get("_meta", {})
                    replayed_from = {"file": "profiles/pmc_traffic.json", "commit": meta.get("commit"), "date": meta.get("date"),
                                     "command": meta.get("command"), "fields": ["roofline.traffic", "roofline.limiter (all but tile_walk)"]}
                except Exception:
                    traffic = None
            # what actually limits the kernel (rocprofv3 SQ counters of this command, tools/profile_round.sh ->
            # profiles/pmc_traffic.json): the blend kernels are bound by VALU issue, not by HBM
            limiter = {}
            try:
                limiter = json.load(open(tpath)).get(dom, {}).get("limiter", {})
            except Exception:
                limiter = {}
            if walk:
                limiter = dict(limiter, tile_walk=walk)
            out["roofline"] = {"bound": limiter.get("bound", "hbm"), "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "replayed_from": replayed_from,
                               "limiter": limiter,
                               "avg_launch_ms": stages[dom]["ms_avg"],
                               "timing": "HIP events on the launch stream, %d extra steps before the warm-up / timed region"
                                         % min(args.steps, 20) +
                                         ("" if args.stacked else " with the frames of a step serialised on one stream"),
                               "frames_per_launch": units, "algorithmic_bytes_per_launch": launch_bytes}
        if world == 1 and args.fit_steps > 0:
            out["fit_step"] = fit_step_rate(dev, N, W, H, args.fit_steps)
            out["fit_step_geometry"] = fit_step_rate(dev, N, W, H, args.fit_steps, start_step=8001)
            if args.fit_densify_steps > 0:
                out["fit_step_densify"] = fit_step_rate(dev, N, W, H, args.fit_densify_steps, start_step=501, densify=True)
            # the reference's default flag value: the networks train too (step >= 12 000: their optimizer steps)
            if args.fit_optim_warp:
              out["fit_step_optim_warp"] = fit_step_rate(dev, N, W, H, args.fit_steps, start_step=12001, optim_warp=True)
              out["fit_step_optim_warp_unfused"] = fit_step_rate(dev, N, W, H, max(10, args.fit_steps // 2), start_step=12001,
                                                                 optim_warp=True, fused_warp_trainable=False, captured=False)
              # (the same steps driven by the eager Python loop of rounds 1-5: what the captured graph replaces)
              out["fit_step_optim_warp_eager"] = fit_step_rate(dev, N, W, H, max(10, args.fit_steps // 2), start_step=12001,
                                                               optim_warp=True, captured=False)
              # (... and the frozen step forced through the captured graph: slower than its eager loop, which is GPU-bound)
              out["fit_step_captured"] = fit_step_rate(dev, N, W, H, max(10, args.fit_steps // 2), captured=True)
        if world == 1:
            # MODELLED multi-GPU figures (no multi-GPU node is reachable from the build box; the driver
            # measures the real curve when it has one): measured 1-GPU step + the all-reduce cost model above
            payload = N * GRAD_FLOATS_PER_SURFEL * 4
            cont = measured_contention()
            sm = {"label": "MODELLED, not measured: measured 1-GPU step + cost model of one all-reduce per step; the only "
                           "measured multi-tenant term is `contention`",
                  "north_star_row": "fit_step.speedup_exchange_serial (a fitting step cannot start before its gradients are "
                                    "reduced; the op-level rows below have no optimizer between steps)",
                  "assumptions": {"xgmi_links_per_gpu": XGMI_LINKS, "GBps_per_link": XGMI_LINK_GBPS,
                                  "latency_ms_per_collective_phase": COLLECTIVE_LATENCY_MS,
                                  "ring_all_links_efficiency": "0.7 -- a guess, not a measurement",
                                  "weak_scaling": "per-GPU work fixed (2 frames per step)"},
                  "contention": cont,
                  "op_level": {"ms_per_step_1gpu": out["ms_per_step"], "payload_bytes": payload,
                               # the exchange of step i beside the kernels of step i + 1 (double-buffered gradients): the
                               # compute slows down by the measured co-tenant factor while a collective is in flight, and
                               # a collective longer than the (slowed) step bounds the rate -- an UPPER BOUND on what
                               # overlapping can give, not a prediction (RCCL's own kernels were never run here)
                               "upper_bound_exchange_hidden": scaling_model(out["ms_per_step"], payload, True, cont["factor"]),
                               "speedup_exchange_serial": scaling_model(out["ms_per_step"], payload, False)}}
            for key in ("fit_step", "fit_step_geometry"):
                if key in out:
                    # Stage3Trainer's exchange: 58 floats per surfel + 3 (learnable background); regist_feat and SH bands
                    # above the active degree stay home (stage3.py: exchanged_params / _packs_rest)
                    pay = N * GRAD_FLOATS_PER_SURFEL * 4 + 12
                    sm[key] = {"ms_per_step_1gpu": out[key]["ms_per_step"], "payload_bytes": pay,
                               "speedup_exchange_serial": scaling_model(out[key]["ms_per_step"], pay, False),
                               # as Stage3Trainer issues it: the SH rest bands' collective behind the rasterizer's
                               # backward, beside the warp's backward (allreduce_gradients)
                               "speedup_rest_bands_beside_the_warp_backward": fit_scaling_model(out[key]["ms_per_step"], N,
                                                                                                cont["factor"]),
                               "warp_backward_ms_assumed": WARP_BACKWARD_MS}
            out["scaling_modelled"] = sm
        if world == 1 and args.host_probe:
            out["host_cost"] = host_cost_probe()
        if world == 1 and args.cpu_images > 0:
            out["cpu_baseline"] = cpu_baseline(scene_cpu, args.cpu_images)
        if world == 1 and args.torch_cpu_images > 0:
            out["cpu_baseline_pytorch"] = torch_cpu_baseline_bounded(args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; flush it so that the JSON is the LAST line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
