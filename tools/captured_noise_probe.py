import sys, os, numpy as np, torch, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
from tests.test_gpu_captured_step import _trainer
from vidu4d_amd.lab4d.stage3 import synthetic_batch
dev = torch.device("cuda:0")
H = W = 96
opts = dict(gs_optim_warp=True, optim_warp_neus_iters=2, iters_per_round=100, num_rounds=1)
def run(captured, steps=14, **extra):
    m, tr = _trainer(dev, captured, **opts, **extra)
    hist = []
    for i in range(steps):
        tr.train_step(synthetic_batch(m, [(2 * i) % 16, (2 * i + 1) % 16], H, W, seed=i))
        hist.append(m._xyz.detach().clone())
    tr.settle(); torch.cuda.synchronize()
    nets = torch.cat([p.detach().reshape(-1) for p in list(m.warp.parameters()) + list(m.camera_mlp.parameters())])
    return hist, nets, dict(tr.captured_stats)
runs = {}
for name, cap, extra in (("e1", False, {}), ("e2", False, {}), ("e3", False, {}), ("c1", True, {}), ("c2", True, {}), ("c3", True, {}),
                         ("e_inline", False, dict(graphed_warp_networks="inline")), ("e_nograph", False, dict(graphed_warp_networks=False))):
    runs[name] = run(cap, **extra)
    print(name, runs[name][2])
for a, b in itertools.combinations(runs, 2):
    per_step = [float((x - y).abs().median()) for x, y in zip(runs[a][0], runs[b][0])]
    print(a, b, "xyz median |diff| per step:", " ".join(f"{v:.1e}" for v in per_step), "| nets max", float((runs[a][1] - runs[b][1]).abs().max()))
