"""Times the gradient-norm (clip) kernel and the fused Adam step alone at the fit step's size (GPU box):
200k surfels x 58 floats in one flat buffer.  Usage: python tools/optim_bench.py [floats]"""
import os, sys
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd.gs.surfel_optim import clip_coef

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000 * 58
dev = torch.device("cuda:0")
g = torch.randn(n, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


us = timed(lambda: clip_coef([g], 5.0))
print(f"clip_coef, one tensor of {n} floats: {us:.1f} us ({n * 4 / us / 1e6:.2f} TB/s)")
parts = list(g.split((n + 6) // 7))
us = timed(lambda: clip_coef(parts, 5.0))
print(f"clip_coef, 7 tensors: {us:.1f} us ({n * 4 / us / 1e6:.2f} TB/s)")
us = timed(lambda: torch.linalg.vector_norm(g))
print(f"torch.linalg.vector_norm: {us:.1f} us")
norm, coef = clip_coef([g], 5.0)
ref = torch.linalg.vector_norm(g.double()).item()
print("norm", float(norm), "reference", ref, "rel", abs(float(norm) - ref) / ref)
