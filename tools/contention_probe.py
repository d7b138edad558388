#!/usr/bin/env python
"""What does a collective's kernel cost the blend kernels it shares the GPU with?  (MEASUREMENT; GPU box.)
No multi-GPU node is reachable from the build box, so the all-reduce of the step's 46 MB gradient buffer cannot be run;
what CAN be measured on one GPU is the contention its kernel causes: a device-to-device copy of the same 46.4 MB held to
16 / 32 workgroups (RCCL's kernels run a fixed number of channels, one workgroup each) loops on a side stream while the
headline step runs, and the stage timers (HIP events on the launch stream) give every kernel's duration with and without
that tenant.  The copy moves HBM bytes at a rate no xGMI link reaches (~1 TB/s against 153 GB/s per link), so this is an
upper bound on the bandwidth side of the interference.
    python tools/contention_probe.py > profiles/r04_contention.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import diff_surfel_rasterization as dsr  # noqa: E402
from vidu4d_amd import _C, _lib  # noqa: E402
from vidu4d_amd.synthetic import frame_motion, make_scene, make_upstream_grads  # noqa: E402

N, W, F = 200_000, 512, 2
dev = torch.device("cuda:0")
scene = make_scene(N, W, None, seed=1234).to(dev)
H = scene.height
dc, do = (t.to(dev) for t in make_upstream_grads(W, H))
frames = [frame_motion(scene, f, 120) for f in range(8)]
rs = dsr.GaussianRasterizationSettings(H, W, scene.tanfovx, scene.tanfovy, scene.bg, 1.0, scene.viewmatrix, scene.projmatrix,
                                       scene.sh_degree, scene.campos, False, False)
dcs, dos = torch.stack([dc] * F, 1).contiguous(), torch.stack([do] * F, 1).contiguous()
PAYLOAD = N * 58 * 4
src = torch.zeros(PAYLOAD // 4, device=dev)
dst = torch.empty_like(src)
side = torch.cuda.Stream(device=dev)
lib = _lib.load()
opac = scene.opacities.clone().requires_grad_(True)
scales = scene.scales.clone().requires_grad_(True)
shs = scene.shs.clone().requires_grad_(True)


def step(k, tenant_wgs):
    ids = [(k * F + i) % 8 for i in range(F)]
    m = torch.stack([frames[i].means3D for i in ids]).requires_grad_(True)
    r = torch.stack([frames[i].rotations for i in ids]).requires_grad_(True)
    with _C.deferred_capacity_check():
        color, radii, allmap = dsr.rasterize_frames(m, torch.zeros_like(m, requires_grad=True), shs, opac, scales, r, [rs] * F)
        if tenant_wgs:
            # the tenant starts when the forward is done (as a collective of the previous step's gradients would still be
            # running) and loops long enough to cover the whole backward
            side.wait_stream(torch.cuda.current_stream(dev))
            _lib.check(lib.vidu4d_diag_copy(dst.data_ptr(), src.data_ptr(), PAYLOAD, tenant_wgs, 12, side.cuda_stream), "diag_copy")
        torch.autograd.backward([color, allmap], [dcs, dos])
    _C.check_deferred()
    for t in (opac, scales, shs):
        t.grad = None


out = {"what": "stage timers (ms per launch of two stacked frames, 200k surfels, 512^2) with a 46.4 MB device-to-device copy "
               "looping on a side stream during the backward, held to N workgroups of 512 threads; 0 = no tenant",
       "payload_bytes": PAYLOAD, "runs": {}}
for wgs in (0, 16, 32, 64, 0):
    for k in range(6):
        step(k, wgs)
    torch.cuda.synchronize()
    _lib.profile_read(reset=True)
    _lib.profile_enable(True)
    for k in range(30):
        step(k, wgs)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    p = {k: round(ms / n, 4) for k, (ms, n) in _lib.profile_read(reset=True).items() if n}
    key = f"{wgs}" if f"{wgs}" not in out["runs"] else f"{wgs} (again)"
    out["runs"][key] = p
base = out["runs"]["0"]
bwd = lambda p: p["blend_bwd"] + p["preprocess_bwd"] + p["bwd_zero"]  # noqa: E731
out["backward_slowdown"] = {k: round(bwd(v) / bwd(base), 4) for k, v in out["runs"].items()}
out["blend_bwd_slowdown"] = {k: round(v["blend_bwd"] / base["blend_bwd"], 4) for k, v in out["runs"].items()}
print(json.dumps(out, indent=1))
