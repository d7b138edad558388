"""Host-side behaviour of the Stage-3 trainer that needs no GPU: the frozen warp table over RAW frame ids, what a
checkpoint load leaves of the networks' schedule, the resumed-run schedule, and the accumulation of network gradients
inside a round (reference: lab4d/engine/trainer.py:37, :268-286, :449, :592-598, :861-869)."""
import numpy as np
import torch

from vidu4d_amd.lab4d import checkpoint as ck
from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
from vidu4d_amd.lab4d.nets import make_frame_info
from vidu4d_amd.lab4d.stage3 import Stage3Trainer


def _model(opts=None, n=60, seed=0, data_info=None, num_frames=6):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    m = DeformableSurfels(dict(fg_motion="gs-bob") | (opts or {}), num_frames=num_frames, device="cpu", data_info=data_info)
    m.init_from_points(rng.normal(size=(n, 3)).astype(np.float32) * 0.2, rng.uniform(size=(n, 3)).astype(np.float32))
    return m


def test_frozen_warp_table_covers_raw_frame_ids():
    """Frame ids are RAW ids (vidloader.stage3_batch: frame_map[idx] + frame_offset_raw[vid]); a video with filtered
    frames has more raw ids than kept frames, and the last ones must not fall off the bone / camera tables."""
    # two videos: 7 of 9 and 5 of 5 frames kept (the layout of tests/golden/dataset_fixture.py)
    kept = [[0, 1, 2, 4, 5, 6, 8], [0, 1, 2, 3, 4]]
    raw_off = np.array([0, 9, 14])
    info = make_frame_info(np.array([0, 7, 12]), raw_off, [i + raw_off[v] for v, ks in enumerate(kept) for i in ks])
    rt = torch.eye(4).repeat(14, 1, 1)
    rt[:, 2, 3] = 3.0
    m = _model(data_info={"frame_info": info, "rtmat": rt.numpy()})
    assert m.num_frames == 12 and int(m.frame_offset_raw[-1]) == 14
    tab = m._frozen_warp_table()
    assert tab["se3_qr"].shape[0] == 14 and tab["cam_q"].shape[0] == 14
    ids = torch.tensor([12, 13, 8])   # the last raw frames of video 1, the last of video 0
    t_art, rest_art = m.warp.articulation.get_vals_and_mean(ids)
    from vidu4d_amd.lab4d import quat_transform as qt
    se3 = qt.dual_quaternion_mul(t_art, qt.dual_quaternion_inverse(rest_art))
    assert torch.allclose(tab["se3_qr"][ids], se3[0], atol=1e-6) and torch.allclose(tab["se3_qd"][ids], se3[1], atol=1e-6)
    cq, ct = m.camera_mlp.get_vals(ids)
    assert torch.allclose(tab["cam_q"][ids], cq, atol=1e-6) and torch.allclose(tab["cam_t"][ids], ct, atol=1e-6)


def test_checkpoint_load_keeps_the_network_schedule(tmp_path):
    """load_checkpoint re-initialises the trainer: the options that size the networks' one-cycle schedule and say when
    their optimizer starts (not among the surfel defaults) must survive, and the run counts as resumed."""
    opts = dict(gs_optim_warp=True, num_rounds=100, iters_per_round=200, optim_warp_neus_iters=5, learning_rate=5e-4)
    a = _model(opts)
    ta = Stage3Trainer(a, opts)
    assert ta.scheduler.total_steps == 20000 and ta.optim_warp_from == 5 and not ta.is_resumed
    # fresh run: warm-up from lr / 25 (trainer.py:272-275)
    first = ta.optimizer.param_groups[0]
    assert np.isclose(first["lr"], first["max_lr"] / 25.0)
    path = ck.save_checkpoint(ta, str(tmp_path), round_count=0)
    b = _model(opts, seed=1)
    tb = Stage3Trainer(b, opts)
    ck.load_checkpoint(path, b, tb)
    assert tb.scheduler.total_steps == 20000 and tb.optim_warp_from == 5 and tb.iters_per_round == 200
    assert tb.is_resumed
    # resumed run: starts at the full rate, decays linearly to lr / 5 (trainer.py:268-271)
    g = tb.optimizer.param_groups[0]
    assert np.isclose(g["lr"], g["max_lr"], rtol=1e-3) and np.isclose(g["min_lr"], g["max_lr"] / 5.0)
    # ... and --load_path alone marks a run as resumed, as upstream's `is_resumed = opts["load_path"] != ""`
    assert Stage3Trainer(_model(opts), opts | {"load_path": "x.pth"}).is_resumed


def test_network_gradients_accumulate_within_a_round():
    """Upstream zeroes the warp / camera gradients when their optimizer steps and at the start of a round, not every
    step: before optim_warp_neus_iters they add up over the steps of a round and are part of the norm check_grad clips
    by.  A parameter autograd never reaches keeps grad = None (AdamW skips it)."""
    opts = dict(gs_optim_warp=True, num_rounds=2, iters_per_round=3, optim_warp_neus_iters=4)
    m = _model(opts)
    tr = Stage3Trainer(m, opts)
    net = tr._net_params
    used, unused = net[0], net[1]

    def fake_step(step, scale):
        tr.begin_gradients()
        loss = (used * scale).sum() + m._xyz.sum() * 1e-3     # `unused` is not in the graph
        loss.backward()
        tr._fold_net_gradients(from_slots=False)
        seen = None if used.grad is None else used.grad.clone()
        tr.finish_step(step)
        return seen

    assert torch.allclose(fake_step(0, 1.0), torch.full_like(used, 1.0))
    assert torch.allclose(fake_step(1, 2.0), torch.full_like(used, 3.0))      # accumulated
    assert unused.grad is None
    assert torch.allclose(fake_step(2, 1.0), torch.full_like(used, 4.0))
    assert torch.allclose(fake_step(3, 5.0), torch.full_like(used, 5.0))      # a new round starts from zero
    before = used.detach().clone()
    assert torch.allclose(fake_step(4, 1.0), torch.full_like(used, 6.0))      # step 4: AdamW steps ...
    assert not torch.equal(before, used.detach())
    assert torch.allclose(fake_step(5, 2.0), torch.full_like(used, 2.0))      # ... and zeroes what it consumed
    assert len(tr.optimizer.state[unused]) == 0                                 # never updated, no weight decay


def test_which_steps_may_be_replayed_from_a_captured_graph():
    """lab4d/captured_step.py replays PLAIN steps only -- the host decisions of the schedule (trainer.py:465-466 SH degree,
    :549-591 densify / prune / opacity reset / outlier pass, :449 and :592-598 the networks' round and their AdamW) stay in the
    eager loop.  The predicate, on the CPU (where nothing is ever captured: there is no CPU rasterizer)."""
    import numpy as np
    from vidu4d_amd.lab4d.deformable_surfels import DeformableSurfels
    from vidu4d_amd.lab4d.stage3 import Stage3Trainer
    rng = np.random.default_rng(0)

    def trainer(**opts):
        m = DeformableSurfels(dict(fg_motion="gs-bob", **opts), num_frames=4, device="cpu")
        m.init_from_points(rng.normal(size=(50, 3)).astype(np.float32) * 0.1, rng.uniform(size=(50, 3)).astype(np.float32))
        return m, Stage3Trainer(m, dict(m.opts))
    m, tr = trainer(captured_step=True)                   # (forced on: also inside the densification regime)
    assert tr.captured_step is False                      # (surfels on the CPU)
    assert not tr._plain_step(0)                          # SH degree 0 -> 1, opacity reset
    assert tr._plain_step(1) and tr._plain_step(499) and tr._plain_step(550)
    assert not tr._plain_step(600)                        # densify (from 500, every 100)
    assert not tr._plain_step(1000) and not tr._plain_step(3000)   # SH raise; opacity reset
    m.active_sh_degree = m.max_sh_degree
    assert tr._plain_step(4001)
    assert not tr._plain_step(4000)                       # densify + the outlier pass (every 2000 from 500)
    assert tr._plain_step(15000) and tr._plain_step(16000)   # past densify_until_iter: nothing but SH, which is at its maximum
    m1, tr1 = trainer()                                   # "auto": the densification regime (steps 500 .. 15000) stays eager
    m1.active_sh_degree = m1.max_sh_degree
    assert tr1._plain_step(499) and not tr1._plain_step(550) and not tr1._plain_step(14999) and tr1._plain_step(15001)
    m2, tr2 = trainer(gs_optim_warp=True, optim_warp_neus_iters=12000, iters_per_round=200, num_rounds=100, captured_step=True)
    m2.active_sh_degree = m2.max_sh_degree
    assert tr2.optimizer is not None
    assert not tr2._plain_step(11999)                     # the networks' gradients still accumulate over the round
    assert not tr2._plain_step(12000) and not tr2._plain_step(12200)   # a round starts: accumulation reset
    assert tr2._plain_step(12001) and not tr2._plain_step(12100)       # ... 12100 densifies
