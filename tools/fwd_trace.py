#!/usr/bin/env python
"""Where does a blend_FWD launch spend its time?  (the forward twin of tools/bwd_trace.py: variant build -DSURFEL_FWD_TRACE,
tools/make_variants.sh "ftrace:-DSURFEL_FWD_TRACE")
Original header of the backward tool follows.
Where does a blend_bwd launch spend its time?  (TEST / MEASUREMENT INFRASTRUCTURE; GPU box.)
Needs the trace variant of the library -- tools/make_full_variant.sh trace "-DSURFEL_BWD_TRACE" -- swapped in for the
product (tools/run_variants.sh does the same): every workgroup of blend_bwd then stores its start, end-of-prologue and
end time (s_memrealtime, 10 ns), its CU / XCD and the list entries it walked.  Prints, for the headline step (two stacked
frames) with recorded segments and with the whole-tile backward: workgroup counts and durations, prologue times, how many
workgroups a CU holds over time, and how much of the launch runs with fewer than four.
    python tools/bwd_trace.py [surfels] [res]
TRACE_OBJECT_RADIUS=r: the object-centric scene (long lists); TRACE_SPLIT=1: the segment-parallel forward's backward
(split_used == 1) instead of the recorded segments."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import diff_surfel_rasterization as dsr  # noqa: E402
from vidu4d_amd import _C, _lib  # noqa: E402
from vidu4d_amd.synthetic import frame_motion, make_object_scene, make_scene, make_upstream_grads  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
F = 2
dev = torch.device("cuda:0")
if os.environ.get("TRACE_OBJECT_RADIUS"):
    kw = {"sigma_px": float(os.environ["TRACE_SIGMA_PX"])} if os.environ.get("TRACE_SIGMA_PX") else {}
    if os.environ.get("TRACE_OPACITY_MODE"):
        kw["opacity_mode"] = os.environ["TRACE_OPACITY_MODE"]   # "init": nothing saturates before the lists end
    scene = make_object_scene(N, W, None, radius=float(os.environ["TRACE_OBJECT_RADIUS"]), seed=1234, **kw).to(dev)
else:
    scene = make_scene(N, W, None, seed=1234).to(dev)
if os.environ.get("TRACE_SPLIT"):
    _C._SPLIT = os.environ["TRACE_SPLIT"]
H = scene.height
AUX = {"alpha": _lib.AUX_ALPHA, "geom": _lib.AUX_GEOM}.get(os.environ.get("TRACE_AUX", ""), 0)   # (TRACE_AUX=alpha / geom: that blend instance)
dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
frames = [frame_motion(scene, f, 120) for f in range(8)]
rs = dsr.GaussianRasterizationSettings(H, W, scene.tanfovx, scene.tanfovy, scene.bg, 1.0, scene.viewmatrix, scene.projmatrix,
                                       scene.sh_degree, scene.campos, False, False)
dcs, dos = torch.stack([dc] * F, 1).contiguous(), torch.stack([do] * F, 1).contiguous()
MAXWG = 1 << 16
import ctypes as C  # noqa: E402
lib = _lib.load()


def step(k):
    ids = [(k * F + i) % 8 for i in range(F)]
    m = torch.stack([frames[i].means3D for i in ids])
    r = torch.stack([frames[i].rotations for i in ids])
    with torch.no_grad():
        dsr.rasterize_frames(m, torch.zeros_like(m), scene.shs, scene.opacities, scene.scales, r, [rs] * F, aux_planes=AUX)


for k in range(5):
    step(k)
torch.cuda.synchronize()
trace = torch.zeros(4 * MAXWG, dtype=torch.int64, device=dev)
assert lib.vidu4d_diag_set_forward_trace(C.c_void_p(trace.data_ptr())) == 0
step(5)
torch.cuda.synchronize()
lib.vidu4d_diag_set_forward_trace(C.c_void_p(0))
t = trace.cpu().numpy().reshape(-1, 4)
t = t[t[:, 0] != 0]
t0, t2 = (t[:, i].astype(np.float64) * 0.01 for i in (0, 2))   # microseconds
start = t0.min()
t0, t2 = t0 - start, t2 - start
hw = t[:, 3]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (((hw >> 32) & 0xf) << 8)
entries = (hw >> 40) & 0xffff
real = entries > 0
dur = t2 - t0
print(f"== blend_fwd: {len(t)} workgroups ran ({int(real.sum())} walked entries), launch span {t2.max():.1f} us, {len(np.unique(cu))} CUs")
print(f"   entries walked per working workgroup: median {int(np.median(entries[real]))}, p90 {int(np.percentile(entries[real], 90))}, max {int(entries.max())}; "
      f"lifetime median {np.median(dur[real]):.1f} us, p90 {np.percentile(dur[real], 90):.1f}, max {dur[real].max():.1f}; us per entry median {np.median(dur[real] / entries[real]):.3f}")
order = np.argsort(t0[real])
q = np.array_split(order, 8)
print("   by start-time octile: start us, entries, lifetime us:", [(round(float(np.median(t0[real][i])), 0), int(np.median(entries[real][i])),
                                                                    round(float(np.median(dur[real][i])), 1)) for i in q])
grid = np.linspace(0, t2.max(), 200)
cus = np.unique(cu[real])
res = np.zeros((len(cus), len(grid)))
idx = {c: i for i, c in enumerate(cus)}
for a_, b_, c_ in zip(t0[real], t2[real], cu[real]):
    res[idx[c_], (grid >= a_) & (grid < b_)] += 1
print("   mean working workgroups per CU over time (10 samples):", [round(float(x), 2) for x in res.mean(0)[::20]])
print(f"   time-averaged: {res.mean():.2f} workgroups per CU; CU-time with < 4: {(res < 4).mean():.2%}, with < 2: {(res < 2).mean():.2%}")
last = np.array([t2[real][cu[real] == c].max() for c in cus])
print(f"   last workgroup of a CU ends at: min {last.min():.0f}, median {np.median(last):.0f}, max {last.max():.0f} us")
