"""Times csrc/lbs.hip's fused skinning kernels alone at the fit step's shape (GPU box): 2 frames, 200k surfels, 25 bones,
bone coordinates from the (3B, 3) bone map, delta-skin logits given.  rocprofv3 --kernel-trace --stats around this script
gives the per-kernel durations (the autograd wrapper adds a few torch kernels).  Usage: python tools/lbs_bench.py [N] [B] [M]"""
import os, sys
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd.lab4d import quat_transform as qt
from vidu4d_amd.lab4d.lbs_fused import lbs_skin_apply

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
M = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
qr = torch.nn.functional.normalize(torch.randn(M, B, 4, generator=g), dim=-1)
se3 = [t.to(dev) for t in qt.quaternion_translation_to_dual_quaternion(qr, 0.3 * torch.randn(M, B, 3, generator=g))]
xyz = (0.5 * torch.randn(N, 3, generator=g)).to(dev).requires_grad_(True)
rot = torch.randn(N, 4, generator=g).to(dev).requires_grad_(True)
raw = torch.randn(B, N, generator=g).to(dev).requires_grad_(True)
cq = torch.nn.functional.normalize(torch.randn(M, 4, generator=g), dim=-1).to(dev)
ct = torch.randn(M, 3, generator=g).to(dev)
A, c = (1.5 * torch.randn(3 * B, 3, generator=g)).to(dev), (0.3 * torch.randn(3 * B, generator=g)).to(dev)
gx, gr = torch.randn(M, N, 3, generator=g).to(dev), torch.randn(M, N, 4, generator=g).to(dev)


def step():
    ox, orot = lbs_skin_apply(None, raw, (se3[0], se3[1]), xyz, rot, cq, ct, unit_rot=True, bone_map=(A, c))
    torch.autograd.backward([ox, orot], [gx, gr])
    xyz.grad = rot.grad = raw.grad = None


for _ in range(3):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record()
torch.cuda.synchronize()
print(f"lbs_skin forward + backward, M={M} N={N} B={B}: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per step (incl. autograd glue)")
