#!/usr/bin/env python
"""Where the HOST time of a rasterizer call goes (no GPU needed): the Python surface run on CPU tensors with the three
launching entry points of the C ABI stubbed out (everything else -- argument marshalling, output allocation, autograd -- is
the product's own code).  The GPU box measures the same path end to end on a scene whose kernels take next to nothing
(profiles/r05_host_probe.txt); this says which lines to look at.  Usage: python tools/host_overhead_probe.py [--profile]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vidu4d_amd import _C, _lib
import diff_surfel_rasterization as dsr
from vidu4d_amd.synthetic import make_scene, make_upstream_grads


class Stub:
    def __init__(self, lib):
        self._lib = lib
    def __getattr__(self, name):
        if name in ("vidu4d_surfel_forward_plan", "vidu4d_surfel_forward_run", "vidu4d_surfel_backward"):
            return lambda *a: 0
        if name == "vidu4d_surfel_num_rendered":
            def count(args, stream, out):
                out._obj.value = 5000
                return 0
            return count
        return getattr(self._lib, name)


class FakeEvent:
    def record(self, *a): pass
    def synchronize(self): pass


def main():
    real = _lib.load()
    _lib._lib = Stub(real)
    _C._check_cuda = lambda *a: None
    _C._stream = lambda dev: 0
    _slot = torch.zeros(16, dtype=torch.int32)
    _C._pinned_slot = lambda dev, ctx: _slot
    _C._cu_count["cpu"] = 256
    torch.cuda.Event = FakeEvent
    torch.cuda.current_stream = lambda dev=None: type("S", (), {"cuda_stream": 0})()
    torch.empty = torch.zeros   # (the header the deferred check reads back must not be garbage; 100 surfels: the fill is free)
    sc = make_scene(100, 64, seed=1)
    dc, do = make_upstream_grads(64, 64)
    rs = dsr.GaussianRasterizationSettings(sc.height, sc.width, sc.tanfovx, sc.tanfovy, sc.bg, 1.0, sc.viewmatrix, sc.projmatrix,
                                           sc.sh_degree, sc.campos, False, False)
    rast = dsr.GaussianRasterizer(rs)
    opac, scales, shs = (t.clone().requires_grad_(True) for t in (sc.opacities, sc.scales, sc.shs))

    def one_frame():
        m = sc.means3D.detach().requires_grad_(True)
        r = sc.rotations.detach().requires_grad_(True)
        m2d = torch.zeros_like(m, requires_grad=True)
        color, radii, allmap = rast(means3D=m, means2D=m2d, opacities=opac, shs=shs, scales=scales, rotations=r)
        torch.autograd.backward([color, allmap], [dc, do])

    def step():
        with _C.deferred_capacity_check():
            for t in (opac, scales, shs):
                t.grad = None
            one_frame(); one_frame()
        _C.check_deferred()

    _C._capacity_hint[(64, 64, "cpu", 1, None)] = 10000
    for _ in range(50):
        step()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    print(f"per-frame surface, 2 frames per step, launches stubbed: {1e3 * dt:.3f} ms per step on this host")
    if "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(300):
            step()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
