"""csrc/optim.hip against torch: SurfelAdam vs torch.optim.Adam (the optimizer the reference builds,
lab4d/engine/trainer.py:240-255), and the device-side densify_and_prune vs GaussianModel.densify_and_prune (itself pinned
against the imported reference by tests/golden/refpy_densify.npz) on a larger random model."""
import copy
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_surfel_adam_matches_torch_adam(gpu_device):
    from vidu4d_amd.gs.surfel_optim import SurfelAdam
    dev = gpu_device
    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 15, 3), (5000, 1), (5000, 2), (5000, 4), (5000, 16), (3,), (1025,)]
    lrs = [5e-5, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3, 2.5e-3, 2.5e-3, 1e-2]
    a = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = SurfelAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(a, lrs))], lr=5e-4, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=5e-4, eps=1e-15)
    for step in range(6):
        for i, (p, q) in enumerate(zip(a, b)):
            if step == 2 and i == 3:
                p.grad = q.grad = None  # a parameter without gradient is skipped; its step count stays behind
                continue
            gr = (torch.randn(p.shape, generator=g) * (10.0 ** (i - 4))).to(dev)
            if step == 4:
                gr[::7] = 0.0  # exact zeros: v stays tiny, the eps = 1e-15 regime
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(), ob.step()
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (i, float((p - q).abs().max()))
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"])
        for k in ("exp_avg", "exp_avg_sq"):  # (entries that cancel to ~0 are compared at the tensor's scale)
            assert torch.allclose(sa[k], sb[k], rtol=5e-6, atol=1e-6 * float(sb[k].abs().max())), (i, k)
    # same state_dict layout as torch's (checkpoint.py stores it as is)
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["eps"] == 1e-15


def _model(dev, n, seed, with_state=True):
    from vidu4d_amd.gs.gaussian_model import GaussianModel
    from vidu4d_amd.gs.surfel_optim import SurfelAdam
    g = torch.Generator(device="cpu").manual_seed(seed)
    gm = GaussianModel(3, device=dev)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    gm._xyz = torch.nn.Parameter(r(n, 3))
    gm._features_dc = torch.nn.Parameter(r(n, 1, 3))
    gm._features_rest = torch.nn.Parameter(r(n, 15, 3))
    gm._opacity = torch.nn.Parameter(r(n, 1) * 3.0 - 2.0)  # sigmoid below 0.005 for some
    gm._scaling = torch.nn.Parameter(r(n, 2) * 1.2 - 4.0)  # around percent_dense * extent = 0.01, a few above 0.1
    gm._rotation = torch.nn.Parameter(r(n, 4))
    gm._regist_feat = torch.nn.Parameter(r(n, 16))
    gm.percent_dense = 0.01
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "regist_feat")
    ps = (gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation, gm._regist_feat)
    gm.optimizer = SurfelAdam([{"params": [p], "lr": 1e-3, "name": nm} for nm, p in zip(names, ps)], lr=5e-4, eps=1e-15)
    if with_state:
        for nm, p in zip(names, ps):
            if nm == "opacity":
                continue  # as after reset_opacity: no Adam state
            gm.optimizer.state[p] = {"step": torch.tensor(7.0), "exp_avg": r(*p.shape), "exp_avg_sq": r(*p.shape).abs()}
    gm.xyz_gradient_accum = (torch.rand(n, 1, generator=g) * 1e-3).to(dev)
    gm.denom = torch.randint(0, 3, (n, 1), generator=g).float().to(dev)  # zeros -> NaN / inf gradients
    gm.max_radii2D = (torch.rand(n, generator=g) * 40).to(dev)
    return gm, names


@pytest.mark.parametrize("screen", [20, None])
def test_device_side_densify_equals_the_python_path(gpu_device, screen):
    from vidu4d_amd.gs.surfel_optim import densify_and_prune_fused
    dev = gpu_device
    n = 20000
    a, names = _model(dev, n, seed=11)
    b, _ = _model(dev, n, seed=11)
    g = a.xyz_gradient_accum / a.denom
    g[g.isnan()] = 0.0
    sel = (g.squeeze(-1) >= 2e-4) & (a.get_scaling.max(dim=1).values > a.percent_dense * 1.0)
    n_sel = int(sel.sum())
    assert 100 < n_sel < n
    stds = torch.cat([a.get_scaling[sel].repeat(2, 1), torch.zeros(2 * n_sel, 1, device=dev)], dim=-1)
    samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=torch.Generator(device=dev).manual_seed(1))
    a.densify_and_prune(2e-4, 0.005, 1.0, screen, samples=samples)
    densify_and_prune_fused(b, 2e-4, 0.005, 1.0, screen, samples=samples)
    assert a._xyz.shape[0] == b._xyz.shape[0] and a._xyz.shape[0] != n
    get = lambda m: (m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m._regist_feat)  # noqa: E731
    for nm, p, q in zip(names, get(a), get(b)):
        if nm in ("xyz", "scaling"):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-6), nm
        else:
            assert torch.equal(p, q), nm
        sa, sb = a.optimizer.state.get(p), b.optimizer.state.get(q)
        assert (sa is None) == (sb is None), nm
        if sa is not None:
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), nm
            assert float(sa["step"]) == float(sb["step"]) == 7.0
        assert b.optimizer.param_groups[names.index(nm)]["params"][0] is q
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_device_side_densify_draws_its_own_samples(gpu_device):
    """Without `samples`: unit normal draws per (copy, source surfel), scaled by the surfel's extent in its plane."""
    from vidu4d_amd.gs.surfel_optim import densify_and_prune_fused
    dev = gpu_device
    b, _ = _model(dev, 20000, seed=5, with_state=False)
    xyz0, sc0 = b._xyz.detach().clone(), b.get_scaling.detach().clone()
    densify_and_prune_fused(b, 2e-4, 0.005, 1.0, 20, generator=torch.Generator(device=dev).manual_seed(9))
    n = b._xyz.shape[0]
    assert n != 20000 and torch.isfinite(b._xyz).all() and b.denom.shape == (n, 1)
    assert len(b.optimizer.state) == 0


def test_surfel_adam_scales_and_zeroes_gradients_in_the_same_pass(gpu_device):
    """grad_scale (the clip coefficient as a device scalar) and zero_grads folded into the one launch == scaling the
    gradients first, stepping, zeroing afterwards."""
    from vidu4d_amd.gs.surfel_optim import SurfelAdam
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in [(3000, 3), (3000, 15, 3), (7,)]]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = SurfelAdam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(a)], eps=1e-15)
    ob = SurfelAdam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(b)], eps=1e-15)
    coef = torch.tensor(0.37, device=dev)
    for _ in range(3):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).to(dev)
            p.grad, q.grad = gr.clone(), gr * coef
        oa.step(grad_scale=coef, zero_grads=True)
        ob.step()
        for p in a:
            assert float(p.grad.abs().max()) == 0.0
    for p, q in zip(a, b):
        assert torch.equal(p, q)


def test_trainer_folds_clip_and_zero_fill_into_adam(gpu_device, monkeypatch):
    """Stage3Trainer's optimizer phase with the clip's scaling and the flat buffer's zero fill inside the Adam launch ==
    with the separate passes, on the SAME gradients (a whole training run is not comparable entry by entry: Adam with
    eps = 1e-15 turns the rounding noise of the blend's atomic sums into +-lr for entries whose gradient is ~0)."""
    import numpy as np
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device

    def make():
        torch.manual_seed(0)
        rng = np.random.default_rng(4)
        m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=8, device=dev)
        m.init_from_points(rng.normal(size=(3000, 3)).astype(np.float32) * 0.25, rng.uniform(size=(3000, 3)).astype(np.float32))
        return m, Stage3Trainer(m)
    m1, t1 = make()
    m2, t2 = make()
    batch = synthetic_batch(m1, [1, 5], 48, 48, seed=2)
    for rep in range(2):
        t1.bind_flat_gradients()
        t1._forward_backward(batch, rep)
        t2.bind_flat_gradients()
        t2._flat.copy_(t1._flat)
        monkeypatch.setattr(Stage3Trainer, "_fold_clip_into_adam", lambda self: self is t1)
        for t in (t1, t2):
            t.clip_gradients(5.0)
            t._optimizer_step(rep)
        assert t1.__dict__.get("_flat_is_zero") is True and float(t1._flat.abs().max()) == 0.0
        for a, b in zip(t1.surfel_params(), t2.surfel_params()):
            assert torch.allclose(a, b, rtol=0, atol=1e-7 * float(b.abs().max()) + 1e-12), float((a - b).abs().max())
        for t in (t1, t2):
            for p in t.exchanged_params():
                p.grad = None


@pytest.mark.parametrize("sizes,scale", [([(200000, 3), (200000, 15, 3), (200000, 1), (3,)], 1e-3), ([(7,)], 50.0),
                                          ([(1023, 5)] * 18, 1.0), ([(4, 4), (0, 3), (5,)], 1e-8)])
def test_clip_coefficient_kernel_matches_torch(gpu_device, sizes, scale):
    """csrc/optim.hip::clip_kernel (one launch: norm over all tensors + the clamp) == torch.nn.utils.clip_grad_norm_
    as Trainer.check_grad calls it (lab4d/engine/trainer.py:861-869); called twice: the workspace counter is left clean."""
    from vidu4d_amd.gs.surfel_optim import clip_coef
    g = torch.Generator().manual_seed(len(sizes))
    grads = [(torch.randn(*s, generator=g) * scale).to(gpu_device) for s in sizes]
    params = [torch.nn.Parameter(torch.zeros_like(t)) for t in grads]
    for p, t in zip(params, grads):
        p.grad = t.clone()
    ref_norm = torch.nn.utils.clip_grad_norm_(params, 5.0)
    ref_coef = min(1.0, 5.0 / (float(ref_norm) + 1e-6))
    for _ in range(2):
        norm, coef = clip_coef(grads, 5.0)
        assert abs(float(norm) - float(ref_norm)) <= 2e-6 * float(ref_norm) + 1e-30
        assert abs(float(coef) - ref_coef) <= 2e-6 * ref_coef
    scaled = [t * coef for t in grads]
    for p, t in zip(params, scaled):
        if t.numel():
            assert torch.allclose(p.grad, t, rtol=1e-5, atol=0)


def test_train_step_on_one_rank_leaves_the_gradients_unbound(gpu_device):
    """One rank, frozen networks: train_step takes the gradients as autograd hands them over (no flat buffer, no
    accumulation passes), clips through the one-launch kernel and steps; a second trainer on the bound path takes the
    same step from the same gradients."""
    import numpy as np
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device

    def make():
        torch.manual_seed(0)
        rng = np.random.default_rng(4)
        m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=8, device=dev)
        m.init_from_points(rng.normal(size=(3000, 3)).astype(np.float32) * 0.25, rng.uniform(size=(3000, 3)).astype(np.float32))
        return m, Stage3Trainer(m)
    m1, t1 = make()
    m2, t2 = make()
    assert not t1._flat_needed()
    batch = synthetic_batch(m1, [1, 5], 48, 48, seed=2)
    t1.begin_gradients()
    assert t1._flat is None
    t1._forward_backward(batch, 0)
    grads = [p.grad for p in t1.exchanged_params()]
    assert sum(g is not None for g in grads) >= 6  # (a parameter nothing reads, regist_feat, gets none: skipped by Adam)
    t2.bind_flat_gradients()
    for p, g_ in zip(t2.exchanged_params(), grads):
        if g_ is not None:
            p.grad.copy_(g_)
    for t in (t1, t2):
        t.clip_gradients(5.0)
        t._optimizer_step(0)
    # (the norm's partial sums are grouped differently over seven arrays than over one: the coefficient may differ by an ulp)
    for a, b in zip(t1.surfel_params(), t2.surfel_params()):
        assert torch.allclose(a, b, rtol=0, atol=1e-7 * float(b.abs().max()) + 1e-12), float((a - b).abs().max())
    # and the public entry point runs on that path
    out = t1.train_step(batch)
    assert t1._flat is None and all(torch.isfinite(v) for v in out.values())


def test_direct_gradient_outputs_into_the_flat_buffer(gpu_device):
    """Frame-parallel path: the canonical parameters' gradients are written by the rasterizer's backward straight into
    the flat exchange buffer (_C.gradient_buffers) and adopted by autograd -- the same gradients autograd allocates
    itself, .grad of every exchanged parameter a view of the buffer afterwards (nothing for the collective to gather),
    also while only the live SH rows are exchanged."""
    import numpy as np
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer, synthetic_batch
    dev = gpu_device

    def make():
        torch.manual_seed(0)
        rng = np.random.default_rng(4)
        m = DeformableSurfels(dict(fg_motion="gs-bob", densify_until_iter=0), num_frames=8, device=dev)
        m.init_from_points(rng.normal(size=(3001, 3)).astype(np.float32) * 0.25, rng.uniform(size=(3001, 3)).astype(np.float32))
        return m, Stage3Trainer(m)
    for degree, world in ((3, 1), (1, 2)):
        m1, t1 = make()
        m2, t2 = make()
        m1.active_sh_degree = m2.active_sh_degree = degree
        t2.world = world                       # (what decides whether only the live SH rows are packed; no collective is run)
        batch = synthetic_batch(m1, [1, 5], 48, 48, seed=2)
        t1.begin_gradients()
        assert t1._flat is None
        t1._forward_backward(batch, 0)
        t2.bind_flat_gradients(direct=True)
        assert set(t2._direct) == {"dL_dsh_dc", "dL_dopacity", "dL_dscales", "dL_dsh_rest"}
        t2._forward_backward(batch, 0)
        flat_ptr = t2._flat.untyped_storage().data_ptr()
        for a, b in zip(t1.exchanged_params(), t2.exchanged_params()):
            assert b.grad is not None and t2._bound(b)
            if b is not m2._features_rest or world == 1:
                assert b.grad.untyped_storage().data_ptr() == flat_ptr
            # (two executions of the blend backward: the order of its float atomics differs, nothing else)
            assert torch.allclose(a.grad, b.grad, rtol=0, atol=5e-6 * float(a.grad.abs().max())), float((a.grad - b.grad).abs().max())
        for t in (t1, t2):
            for p in t.surfel_params():
                p.grad = None


def test_network_adamw_equals_torch_adamw(gpu_device):
    """gs/surfel_optim.NetworkAdamW (csrc/optim.hip adam_kernel with the decoupled decay, 32 tensors per launch) against
    torch.optim.AdamW -- the reference's optimizer of the warp / camera networks (lab4d/engine/trainer.py:177-286): 70
    tensors of odd sizes in two groups of different rates, one of them never given a gradient, rates changing per step as
    the one-cycle schedule changes them, a clip coefficient riding along.  Parameters and both moments after 12 steps equal
    to rounding (the two evaluate the bias corrections in double / in fp32)."""
    from vidu4d_amd.gs.surfel_optim import NetworkAdamW
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 1 + (i * 7) % 90) if i % 3 else (1 + i,) for i in range(70)]
    init = [torch.randn(*s, generator=g) for s in shapes]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        groups = [{"params": ps[:40], "lr": 5e-4}, {"params": ps[40:], "lr": 5e-3}]
        return ps, cls(groups, lr=5e-4, betas=(0.9, 0.999), weight_decay=1e-4, **kw)

    pa, oa = make(NetworkAdamW)
    pb, ob = make(torch.optim.AdamW)
    for step in range(12):
        coef = torch.tensor(1.0 if step % 2 else 0.37, device=dev)
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k == 13:
                continue   # (never touched by autograd: skipped, weight decay included)
            gr = torch.randn(a.shape, generator=g).to(dev) * (10.0 ** ((k % 5) - 3))
            a.grad = gr.clone()
            b.grad = gr * coef          # (torch's optimizer sees the clipped gradient; ours multiplies on the way in)
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a["lr"] = grp_b["lr"] = grp_b["lr"] * (1.1 if step < 5 else 0.8)
        oa.step(grad_scale=coef)
        ob.step()
    assert len(oa.state[pa[13]]) == 0 and torch.equal(pa[13], init[13].to(dev))
    for k, (a, b) in enumerate(zip(pa, pb)):
        if k == 13:
            continue
        assert float(oa.state[a]["step"]) == 12.0
        for x, y, what in ((a, b, "param"), (oa.state[a]["exp_avg"], ob.state[b]["exp_avg"], "exp_avg"),
                           (oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], "exp_avg_sq")):
            err = float((x - y).abs().max())
            assert err <= 2e-6 * float(y.abs().max()) + 1e-12, (k, what, err, float(y.abs().max()))
    # the checkpoint format is torch's: a state written by torch's AdamW (tensor step counts) loads and steps
    oc_params, oc = make(NetworkAdamW)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))   # (load_state_dict keeps tensors that already match: the two would share moments)
    for a, b in zip(oc_params, pb):
        a.data.copy_(b.data)
        a.grad = torch.ones_like(a)
        b.grad = torch.ones_like(b)
    oc.step()
    ob.step()
    for k, (a, b) in enumerate(zip(oc_params, pb)):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12, k
