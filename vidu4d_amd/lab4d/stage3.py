"""Stage-3 fitting loop (`lab4d/train.py --fg_motion gs-bob`): losses on the rasterizer outputs, the
surfel optimizer, the SH-degree / densify / prune / opacity-reset cadence, and frame-parallel
multi-GPU execution.

Reference: Trainer.optimizer_init (lab4d/engine/trainer.py:240-255: Adam, one group per surfel
tensor, eps 1e-15), Trainer.train_one_round (:439-602), check_grad (:861-884, clip 5.0),
dvr_model.compute_loss (lab4d/engine/model.py:549-584; rgb L1 under vis2d :674-693, mask :648-650,
normal/dist regularisers gated to step > 8000 :817-842, masking :895-960, weighting :980-1012).

Multi-GPU (new relative to upstream, whose DDP wrapper cannot track the re-created surfel parameters,
SURVEY.md §2a): one process per GPU, canonical surfels + warp replicated, every rank renders ITS
frames of the step, then ONE all-reduce over the flat surfel-gradient buffer (RCCL over xGMI with
backend "nccl"; gloo on CPU for tests).  Densification statistics are all-reduced when they are
consumed (sum, sum, max) and the densify / prune decision is replayed identically on every rank from
a step-seeded generator, so the replicas never diverge."""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist

from .deformable_surfels import DeformableSurfels, _Args

SURFEL_GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "regist_feat")


def mask_balance_weight(maskfg, vis2d, is_detected):
    """get_mask_balance_wt (model.py:586-611): foreground and background pixels of the visible, detected
    frames contribute equally to the silhouette loss.  Sync-free (sums and torch.where, no indexing)."""
    vis = vis2d * is_detected
    seen = (vis > 0).to(maskfg.dtype)
    total = vis.sum()
    pos_wt = total / (maskfg * seen).sum()
    neg_wt = total / ((1 - maskfg) * seen).sum()
    both = (maskfg.sum() > 0) & ((1 - maskfg).sum() > 0)
    return torch.where(both, 0.5 * pos_wt * maskfg + 0.5 * neg_wt * (1 - maskfg), torch.ones_like(maskfg))


def mean_of_positive(v):
    """apply_loss_weights (model.py:996-999): mean over the entries > 0, over everything when there is none;
    without boolean-mask indexing (a nonzero() would block the host)."""
    if v.dim() == 0:
        return v
    pos = v > 0
    n = pos.sum()
    return torch.where(n > 0, (v * pos).sum() / n.clamp_min(1), v.mean())


def compute_losses(rendered: dict, batch: dict, step: int, cfg) -> dict:
    """rendered: maps (M,H,W,C) from DeformableSurfels.render_frames; batch: rgb (M,H,W,3), mask (M,H,W,1),
    vis2d (M,H,W,1), is_detected (M,) (optional: all detected).  Returns the weighted scalar terms the reference
    keeps under --rgb_loss_only (trainer.py:477-483): rgb, mask, normal_loss, dist_loss -- the values of
    dvr_model.compute_recon_loss / mask_losses / compute_reg_loss / apply_loss_weights for field_type "fg"
    (model.py:613-693, :895-978, :803-842, :980-1012), pinned in tests/test_refpy_host.py::test_stage3_losses.
    Upstream quirks kept on purpose: the colour term is ONE scalar L1 (then spread over the fg-and-visible
    pixels and averaged back, i.e. it vanishes when no such pixel exists), and the normal-consistency dot
    product is summed over the FRAME axis of the (M,H,W,3) maps (:831: `.sum(dim=0)`)."""
    vis2d = batch["vis2d"].float()
    maskfg = batch["mask"].float()
    M = vis2d.shape[0]
    det = batch["is_detected"].float() if "is_detected" in batch else torch.ones(M, device=vis2d.device)
    det = det.view(M, 1, 1, 1)
    zero = torch.zeros((), device=vis2d.device)
    sel = vis2d.expand(-1, -1, -1, 3) > 0
    l1 = torch.where(sel, torch.abs(rendered["rendered"] - batch["rgb"]), zero).mean() * (1.0 - cfg.lambda_dssim)
    rgb = l1 * (maskfg * vis2d)                                   # mask_losses: type-specific key, fg field
    mask = (rendered["mask"] - maskfg).pow(2) * mask_balance_weight(maskfg, vis2d, det) * vis2d * det
    out = {"rgb": mean_of_positive(rgb) * cfg.rgb_wt, "mask": mean_of_positive(mask) * cfg.mask_wt}
    lam_n = cfg.lambda_normal if step > 8000 else 0.0
    lam_d = cfg.lambda_dist if step > 8000 else 0.0
    # A regulariser with weight 0 (both, for the first 8000 steps, model.py:817-842) contributes a
    # gradient of exactly 0: it is not put on the autograd tape at all, which spares the backward of the
    # whole depth-to-normal chain (~100 launches per step).
    if lam_n != 0.0:
        out["normal_loss"] = lam_n * (1 - (rendered["rend_normal"] * rendered["surf_normal"]).sum(dim=0)).mean()
    else:
        out["normal_loss"] = zero
    out["dist_loss"] = lam_d * rendered["rend_dist"].mean() if lam_d != 0.0 else zero
    return out


class Stage3Trainer:
    def __init__(self, model: DeformableSurfels, opts: dict | None = None, is_resumed: bool | None = None):
        self.model = model
        # the FULL option dict is kept (checkpoint.load_checkpoint re-initialises from it: num_rounds, iters_per_round,
        # optim_warp_neus_iters are not among _Args' defaults)
        self.opts = dict(opts or model.opts)
        self.cfg = _Args(self.opts)
        # trainer.py:37: a run that starts from a checkpoint (Stage-3 on a Stage-2 checkpoint always does)
        self.is_resumed = bool(self.opts.get("load_path", "")) if is_resumed is None else bool(is_resumed)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.current_steps = 0
        c = self.cfg
        m = model
        groups = [
            # (the trainer's own spatial_lr_scale is 1, trainer.py:236 -- not the model's scene radius)
            {"params": [m._xyz], "lr": c.position_lr_init * 1.0, "name": "xyz"},
            {"params": [m._features_dc], "lr": c.feature_lr, "name": "f_dc"},
            {"params": [m._features_rest], "lr": c.feature_lr / 20.0, "name": "f_rest"},
            {"params": [m._opacity], "lr": c.opacity_lr, "name": "opacity"},
            {"params": [m._scaling], "lr": c.scaling_lr, "name": "scaling"},
            {"params": [m._rotation], "lr": c.rotation_lr, "name": "rotation"},
            {"params": [m._regist_feat], "lr": c.feature_lr, "name": "regist_feat"},
        ]
        if c.gs_learnable_bg:
            groups.append({"params": [m.learnable_bkgd], "lr": c.feature_lr, "name": "bg_rgb"})
        # on the GPU: ONE launch per step for all groups (csrc/optim.hip; same update rule and state layout as the
        # reference's torch.optim.Adam)
        if m._xyz.is_cuda:
            from ..gs.surfel_optim import SurfelAdam
            self.gs_optimizer = SurfelAdam(groups, lr=c.learning_rate, eps=1e-15)
        else:
            self.gs_optimizer = torch.optim.Adam(groups, lr=c.learning_rate, eps=1e-15)
        m.optimizer = self.gs_optimizer
        self._flat = None
        self._pending_reduce = None
        self._rest_slot = None      # (N, k, 3) view of the flat buffer while only k < 15 SH rest rows are exchanged
        self._net_slots = []        # the networks' places in the flat buffer (gs_optim_warp=True)
        self._chunk_split = None    # where the second collective (the SH rest bands) starts
        self._net_split = None      # ... and the third (the networks' gradients, when they train)
        self._rest_ready = None     # event: the rasterizer's backward has written the SH gradients (this step)
        self._side_stream = None
        # --gs_optim_warp=False (the README's Stage-3 command): warp and camera networks come from the
        # Stage-2 checkpoint and are never stepped (trainer.py:592-598).  Upstream still back-propagates
        # into them; freezing them is results-equivalent for the surfels up to the gradient clip (upstream's
        # check_grad, :861-869, puts the unused warp gradients into the norm it clips by -- DESIGN.md §5) and skips
        # the weight-gradient GEMMs and the (M,N,B,.) broadcast reductions of the warp backward.
        o = self.opts
        self.optim_warp = bool(o.get("gs_optim_warp", False))
        self.optim_warp_from = int(o.get("optim_warp_neus_iters", 12000))
        net_params = [(n, prm) for mod_name, mod in (("warp", m.warp), ("camera_mlp", m.camera_mlp))
                      for n, prm in ((f"{mod_name}.{k}", v) for k, v in mod.named_parameters())]
        for _, prm in net_params:
            prm.requires_grad_(self.optim_warp)
        self.optimizer = self.scheduler = None
        if self.optim_warp:
            # the reference's second optimizer (trainer.py:177-286): AdamW(betas (0.9, 0.999), weight decay 1e-4), one
            # group per tensor, 10x the base rate for the explicit parameters, linear one-cycle schedule; it is stepped
            # once the surfels have had optim_warp_neus_iters steps (:592-598)
            explicit = (".logibeta", ".logsigma", ".logscale", ".log_gauss", ".base_quat", ".shift")
            # Upstream builds one group per tensor; there are only TWO distinct rates, and AdamW treats every tensor on its
            # own, so one group per rate is the same arithmetic -- and two multi-tensor launches per step instead of 64
            # (round 5: 1.9 ms of host time and 1.3 ms of GPU time per step with training networks;
            # `network_param_groups: "per_tensor"` restores upstream's layout)
            per_tensor = o.get("network_param_groups", "per_rate") == "per_tensor"
            groups, lrs = [], []
            by_rate = {}
            for n, prm in net_params:
                rate = c.learning_rate * (10.0 if any(n.endswith(e[1:]) or e in n for e in explicit) else 1.0)
                if per_tensor:
                    groups.append({"params": [prm], "name": n})
                    lrs.append(rate)
                else:
                    by_rate.setdefault(rate, []).append((n, prm))
            for rate, members in by_rate.items():
                groups.append({"params": [prm for _, prm in members], "name": [n for n, _ in members]})
                lrs.append(rate)
            total = max(2, int(o.get("num_rounds", 1)) * int(o.get("iters_per_round", 200)))
            # (one group per tensor is the reference's layout, trainer.py:240-255; torch's default for-each implementation
            # then issues six launches per GROUP -- ~600 per step for the bob networks, 3.9 ms of GPU time at 6.5 us each,
            # 40 % of the whole step with training networks (tools/fit_optim_warp_profile.py).  `fused=True` is one
            # multi-tensor launch per group and the same arithmetic.)
            on_gpu = all(prm.is_cuda for _, prm in net_params)
            fused = on_gpu and o.get("fused_network_adamw", True)
            # (captured steps, lab4d/captured_step.py: the step counters and the learning rates live on the device -- torch's
            # capturable form of the same update; the scheduler below fills the rate tensors)
            # (whether or not steps end up being replayed from a graph: torch's capturable form evaluates the bias corrections
            # on the device in fp32, the other one on the host in double -- 1e-7 apart per step, which AdamW's normalised
            # updates amplify; one form for both keeps the captured and the eager loop on ONE trajectory)
            capt = fused and self.world == 1 and bool(o.get("capturable_network_adamw", True))
            # Round 6: the same update from the surfel optimizer's kernel (gs/surfel_optim.NetworkAdamW: 32 tensors per
            # launch -- three launches for the 66 tensors where torch's fused capturable form takes seven of ~24 us each,
            # 145 us of a 2.6 ms step; scalars by value, or from device rows when the step is a captured graph's).
            # `network_adamw: "torch"` restores torch's optimizer.
            if on_gpu and o.get("network_adamw", "hip") == "hip":
                from ..gs.surfel_optim import NetworkAdamW
                self.optimizer = NetworkAdamW(groups, lr=c.learning_rate, betas=(0.9, 0.999), weight_decay=1e-4)
            else:
                self.optimizer = torch.optim.AdamW(groups, lr=torch.tensor(float(c.learning_rate), device=m._xyz.device) if capt
                                                   else c.learning_rate, betas=(0.9, 0.999), weight_decay=1e-4,
                                                   **({"fused": True} if fused else {}), **({"capturable": True} if capt else {}))
            # trainer.py:268-275: a resumed run starts the networks at the full rate and decays to lr / 5; a fresh one
            # warms up from lr / 25 over two rounds
            if self.is_resumed:
                div_factor, final_div_factor, pct_start = 1.0, 5.0, 0.0
            else:
                div_factor, final_div_factor = 25.0, 1.0
                pct_start = min(0.5, 2.0 / max(1, int(o.get("num_rounds", 1))))
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
                self.optimizer, lrs, total, pct_start=pct_start, cycle_momentum=False, anneal_strategy="linear",
                div_factor=div_factor, final_div_factor=final_div_factor)
        self._net_params = [prm for _, prm in net_params]
        self.iters_per_round = max(1, int(o.get("iters_per_round", 200)))
        # Upstream zeroes the networks' gradients only when their optimizer steps and at the start of a round
        # (trainer.py:449, :592-598): until optim_warp_neus_iters they ACCUMULATE over the steps of a round and enter
        # check_grad's clip norm (:861-869), which scales the surfel gradients too.  Reproduced: the per-step network
        # gradients (after the exchange) are added to these buffers; a parameter autograd never touched keeps
        # grad = None (AdamW then skips it, weight decay included).
        self._net_accum = [None] * len(self._net_params)
        # plain steps as one captured hipGraph (lab4d/captured_step.py; `captured_step: False` restores the eager loop)
        # "auto" (default): where it pays -- networks that TRAIN, whose eager step is bound by ~2.3 ms of Python (measured, same
        # box, 200 k surfels / 512^2: 3.01 -> 2.65 ms per step); with frozen networks the eager step is GPU-bound already and
        # a replayed graph costs ~1.5 us more per node on this runtime (1.10 -> 1.19 ms): eager stays (profiles/r06_graph_env_ab.txt)
        want = o.get("captured_step", "auto")
        # ("auto" also keeps the densification regime eager: every densify / prune re-creates the surfel tensors, i.e. a new
        # capture (14.6 ms each at 200 k surfels) every 100 steps plus a taken-back step whenever the pair count has outgrown
        # the captured buffers -- measured with training networks, tools/fit_optim_warp_densify_ab.py: 3.33 ms per step
        # captured against 2.59 eager.  In the reference's schedule the networks' optimizer starts at step 12 000 and
        # densification ends at 15 000: the captured steps are the ones behind that.)
        self._capture_while_densifying = want is True
        want = self.optimizer is not None if want == "auto" else bool(want)
        self.captured_step = want and m._xyz.is_cuda and self.world == 1
        self._captured, self._cap_stream, self._inflight, self._streak = {}, None, None, (None, 0)
        self.capture_after = int(o.get("capture_after", 3))   # eager steps of a shape before it is captured (hints settle)
        self.captured_stats = {"captures": 0, "replays": 0, "taken_back": 0}
        # the outlier pass of trainer.py:573-588 (open3d remove_radius_outlier(nb_points=20, radius=0.004))
        self.outlier_radius, self.outlier_nb_points = float(o.get("outlier_radius", 0.004)), int(o.get("outlier_nb_points", 20))
        self.outlier_neighbor_count = None   # (None: csrc/knn.hip on the GPU, no pass for surfels on the CPU)

    # ---- the path's only exchange
    def surfel_params(self):
        m = self.model
        ps = [m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m._regist_feat]
        if self.cfg.gs_learnable_bg:
            ps.append(m.learnable_bkgd)
        return ps

    def live_sh_rows(self) -> int:
        """Rows of `_features_rest` (N, 15, 3) that can receive a gradient: the bands up to the active SH degree
        (forward.cu:20-71 reads (deg + 1)^2 coefficients; the backward writes zeros above them)."""
        m = self.model
        return min(int(m._features_rest.shape[1]), (int(m.active_sh_degree) + 1) ** 2 - 1)

    def exchanged_params(self):
        """Everything whose gradient the ranks must agree on: the surfel tensors a loss of this path reaches -- NOT
        `_regist_feat` (only the feature-matching losses --rgb_loss_only drops read it: its gradient is None on every
        rank and the optimizer skips it) -- the small ones first, the SH rest bands last (the two chunks of the
        exchange), and the networks when they train."""
        m = self.model
        ps = [m._xyz, m._features_dc, m._opacity, m._scaling, m._rotation]
        if self.cfg.gs_learnable_bg:
            ps.append(m.learnable_bkgd)
        ps.append(m._features_rest)
        return ps + (self._net_params if self.optim_warp else [])

    def _packs_rest(self) -> bool:
        """More than one rank and SH bands above the active degree: their gradients are exact zeros on every rank
        (no kernel writes them), so only the live rows [:, :k] cross the wire -- packed into the flat buffer before the
        collective, unpacked after it.  Results-identical; at degree 0 (the first 1000 steps) 45 of 58 floats per
        surfel stay home."""
        return self.world > 1 and self.live_sh_rows() < int(self.model._features_rest.shape[1])

    def bind_flat_gradients(self, direct: bool = False):
        """Makes every exchanged parameter's .grad a view of ONE persistent fp32 buffer and zeroes it (this is
        the step's zero_grad): autograd then accumulates straight into the buffer the all-reduce, the norm for
        the clip and the fused Adam read -- no gather / scatter copies around the collective.  Re-bound every
        step because densify / prune re-create the surfel parameters.  Layout: [xyz | f_dc | opacity | scaling |
        rotation | bg] (chunk 0) [f_rest (live rows only while the SH degree is below 3) | networks] (chunk 1).
        direct (what begin_gradients asks for): the canonical parameters a stacked rasterizer call differentiates are NOT
        pre-bound -- their slices are handed to the rasterizer's backward as output buffers (_gradient_outputs) and
        become their .grad when autograd adopts what it returns."""
        ps = self.exchanged_params()
        rest = self.model._features_rest
        pack = self._packs_rest()
        k = self.live_sh_rows()
        net_ids = {id(p) for p in self._net_params} if self.optim_warp else set()
        sizes = [(rest.shape[0] * k * rest.shape[2]) if (p is rest and pack) else p.numel() for p in ps]
        # every tensor starts on a 256-byte boundary (the rasterizer's backward writes some of them in place with 16-byte
        # stores, _C.gradient_buffers); the few padding floats stay zero
        starts, n = [], 0
        for sz in sizes:
            starts.append(n)
            n += (sz + 63) // 64 * 64
        # tensors whose gradient the rasterizer's backward produces directly and exactly once (canonical parameters through
        # one stacked call): handed over as output buffers -- no zero fill, no accumulation pass (self._direct)
        m = self.model
        direct_ok = (direct and ps[0].is_cuda and m.opts.get("stacked_frames", True) and m.opts.get("canonical_params", True)
                     and m.opts.get("fused_loss", True) and m.opts.get("direct_gradient_outputs", True)
                     and m._features_rest.shape[1] == 15)
        self._direct = {}
        if self._flat is None or self._flat.numel() != n or self._flat.device != ps[0].device:
            self._flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            fresh = True
        else:
            fresh = self.__dict__.pop("_flat_is_zero", False)  # (the one-launch Adam left every gradient zero-filled)
            if not fresh:
                self._flat.zero_()
        self.__dict__.pop("_flat_is_zero", None)
        self._rest_slot = None
        self._net_slots = []
        self._chunk_split = None
        self._net_split = None
        self._rest_ready = None
        names = {id(m._features_dc): "dL_dsh_dc", id(m._opacity): "dL_dopacity", id(m._scaling): "dL_dscales"}
        if not pack:
            names[id(rest)] = "dL_dsh_rest"
        for p, sz, off in zip(ps, sizes, starts):
            if p is rest:
                self._chunk_split = off
            if p is rest and pack:
                # the full-size gradient the kernels write lives outside the flat buffer; its live rows are copied in
                # and out around the collective
                full = self.__dict__.get("_rest_full")
                if full is None or full.shape != rest.shape or full.device != rest.device:
                    full = self._rest_full = torch.zeros_like(rest)
                elif not fresh and not direct_ok:
                    full.zero_()
                if direct_ok:   # (written whole by the backward, dead bands as zeros)
                    p.grad = None
                    self._direct["dL_dsh_rest"] = full
                else:
                    p.grad = full
                self._rest_slot = self._flat[off:off + sz].view(rest.shape[0], k, rest.shape[2])
            elif id(p) in net_ids:
                if self._net_split is None:
                    self._net_split = off
                # fresh per-step gradient (None when autograd never reaches the parameter); it is packed into the
                # buffer for the collective and then ADDED to the round's accumulated gradient (_fold_net_gradients)
                p.grad = None
                self._net_slots.append(self._flat[off:off + sz].view_as(p))
            elif direct_ok and id(p) in names:
                p.grad = None                      # (autograd adopts the tensor the backward returns: a view of the buffer)
                self._direct[names[id(p)]] = self._flat[off:off + sz].view_as(p)
            else:
                p.grad = self._flat[off:off + sz].view_as(p)
        return self._flat

    def _gradient_outputs(self):
        """Context for the step's backward: the flat buffer's slices of the canonical parameters as the rasterizer's
        gradient outputs (bind_flat_gradients decided which)."""
        from .. import _C
        if self._flat is None:
            return contextlib.nullcontext()
        direct = self.__dict__.get("_direct") or {}
        # more than one rank: note when the SH gradients -- 45 of the 58 floats per surfel -- are written, so that their
        # collective can start then instead of after the warp's backward (allreduce_gradients)
        early = self._note_rest_written if (self.world > 1 and "dL_dsh_rest" in direct and
                                            self.model.opts.get("early_exchange", True)) else None
        # (on the MODEL's rasterizer context: that is the one its forwards run under -- render_frames -- and their backward reads)
        return _C.gradient_buffers(on_written=early, context=self.model.raster_context, **direct)

    def _note_rest_written(self):
        self._rest_ready = torch.cuda.Event()
        self._rest_ready.record(torch.cuda.current_stream(self._flat.device))

    def _flat_needed(self):
        """One rank, surfels on the GPU, frozen networks: nothing reads the gradients but the one-launch clip and the
        one-launch Adam, which take per-tensor pointers -- autograd then hands every parameter its gradient tensor as it
        is (no accumulation into a pre-bound buffer: 16 read-add-write passes per step) and there is nothing to zero."""
        return not (self.world == 1 and self._fold_clip_into_adam())

    def begin_gradients(self):
        """The step's zero_grad: the flat buffer where it is needed (bind_flat_gradients), else plain `grad = None`."""
        if self.optim_warp and self.current_steps % self.iters_per_round == 0:
            self._net_accum = [None] * len(self._net_params)   # trainer.py:449: zero_grad at the start of a round
        if self._flat_needed():
            return self.bind_flat_gradients(direct=True)
        self._flat = None
        for p in self.exchanged_params():
            p.grad = None
        return None

    def _bound(self, p) -> bool:
        if p is self.model._features_rest and self._rest_slot is not None:
            full = self.__dict__.get("_rest_full")
            return p.grad is not None and full is not None and \
                p.grad.untyped_storage().data_ptr() == full.untyped_storage().data_ptr()
        if self.optim_warp and any(p is q for q in self._net_params):
            return True  # (packed by hand below)
        return p.grad is not None and p.grad.untyped_storage().data_ptr() == self._flat.untyped_storage().data_ptr()

    def allreduce_gradients(self, async_op: bool = False):
        """Sum of the flat gradient buffer over the ranks, issued as up to THREE collectives (the SH rest bands, the small
        tensors, the networks): `wait_gradients` joins them one after the other, so that the norm of one chunk is taken
        while the next is still on the wire.  The mean (/ world) is folded into the clip coefficient where the one-launch
        Adam applies it, else applied in wait_gradients.  With async_op the collectives are left in flight (RCCL runs
        them on its own stream) and work that does not read the gradients -- the densification statistics -- overlaps
        them."""
        if self.world == 1:
            return
        if self._flat is None or not all(self._bound(p) for p in self.exchanged_params()):
            # gradients that were not produced into the flat buffer (a caller that set them by hand): gather them
            ps = self.exchanged_params()
            nets = {id(p) for p in self._net_params} if self.optim_warp else set()
            # (a PARTIALLY bound state is reachable: with direct gradient outputs some parameters' .grad are views of the
            # buffers while autograd handed others a tensor of its own -- re-binding zero-fills the buffers, so whatever
            # lives in them is copied out first)
            homes = {t.untyped_storage().data_ptr() for t in (self._flat, self.__dict__.get("_rest_full")) if t is not None}
            grads = [(p.grad.clone() if p.grad is not None and p.grad.untyped_storage().data_ptr() in homes else p.grad)
                     for p in ps]
            self.bind_flat_gradients()
            for p, g_ in zip(ps, grads):
                if id(p) in nets:
                    p.grad = g_
                elif g_ is not None:
                    p.grad.copy_(g_)
        for p, slot in zip(self._net_params if self.optim_warp else [], self._net_slots):
            if p.grad is not None:
                slot.copy_(p.grad)
        split = self._chunk_split if self._chunk_split else 0
        nets = self._net_split if self._net_split else self._flat.numel()
        self._pending_reduce = []
        rest = self._flat[split:nets]
        ready, self._rest_ready = self._rest_ready, None
        if ready is not None and rest.numel():
            # the SH rest bands were written by the rasterizer's backward, before the warp's: their collective waits for
            # that event only (a side stream that has seen nothing else), i.e. it runs while the compute stream is still in
            # the warp's backward and the statistics.  Issued here, after the step's capacity check, so that every rank
            # issues the same collectives whether or not it had to replay its step.
            dev = self._flat.device
            if self._side_stream is None or self._side_stream.device != dev:
                self._side_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._side_stream):
                self._side_stream.wait_event(ready)
                if self._rest_slot is not None and self._rest_slot.numel():
                    self._rest_slot.copy_(self.model._features_rest.grad[:, :self._rest_slot.shape[1]])
                self._pending_reduce.append((rest, dist.all_reduce(rest, op=dist.ReduceOp.SUM, async_op=True)))
            rest = None
        elif self._rest_slot is not None and self._rest_slot.numel():
            self._rest_slot.copy_(self.model._features_rest.grad[:, :self._rest_slot.shape[1]])
        chunks = [c for c in (self._flat[:split], rest, self._flat[nets:]) if c is not None and c.numel()]
        self._pending_reduce += [(c, dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True)) for c in chunks]
        if not async_op:
            self.wait_gradients()

    def wait_gradients(self):
        """Joins the exchange.  Leaves the MEAN over the ranks in the gradients -- or, where the one-launch Adam
        consumes them (_fold_clip_into_adam), the SUM together with `_mean_scale = 1 / world`, which clip_gradients
        folds into the coefficient (one pass over 46 MB less) -- and the squared norm of every chunk, taken as the
        chunk arrives."""
        if self._pending_reduce is None:
            return
        fold = self._fold_clip_into_adam()
        self._chunk_sq = []
        for chunk, work in self._pending_reduce:
            work.wait()
            if fold:
                from ..gs.surfel_optim import clip_coef
                self._chunk_sq.append(clip_coef([chunk], 1.0)[0].square())
            else:
                chunk.div_(self.world)
        self._pending_reduce = None
        self._mean_scale = 1.0 / self.world if fold else 1.0
        if self._rest_slot is not None and self._rest_slot.numel():
            self.model._features_rest.grad[:, :self._rest_slot.shape[1]].copy_(self._rest_slot)
        self._fold_net_gradients(from_slots=True)

    def _fold_net_gradients(self, from_slots: bool, adopt: bool = False):
        """Adds this step's (exchanged) network gradients to the round's accumulated ones and makes those the
        parameters' .grad -- what upstream's never-zeroed .grad holds when check_grad and, later, AdamW read it.
        adopt: the networks' optimizer steps at the end of THIS step and empties the accumulation (_optimizer_step), so a
        first contribution is taken as it is instead of copied (66 tensors: 66 device copies per step; the tensor may be a
        captured graph's static buffer, lab4d/net_graphs.py, which the next step's backward overwrites -- by then unused)."""
        if not self.optim_warp:
            return
        if adopt and not from_slots:
            for i, p in enumerate(self._net_params):
                g = p.grad
                if g is not None and self._net_accum[i] is not None:
                    g = self._net_accum[i].add_(g)
                if g is not None:
                    self._net_accum[i] = g
                p.grad = self._net_accum[i]
            return
        acc, new = [], []   # (the in-place additions of all tensors as ONE for-each call: 66 launches -> a few, every step of
        for i, p in enumerate(self._net_params):   # the 12 000 before AdamW starts)
            if p.grad is None:
                g = None
            elif from_slots:
                g = self._net_slots[i]
            else:
                g = p.grad
            if g is not None:
                if self._net_accum[i] is None:
                    self._net_accum[i] = g.detach().clone()
                else:
                    acc.append(self._net_accum[i])
                    new.append(g.detach())
            p.grad = self._net_accum[i]
        if acc:
            torch._foreach_add_(acc, new)

    def _adamw_steps(self, step: int) -> bool:
        return self.optimizer is not None and step >= self.optim_warp_from

    def _fold_clip_into_both(self, step: int) -> bool:
        """Networks that train, AdamW stepping THIS step (round 6): the gradients are consumed by the two optimizers right
        behind the clip and discarded (trainer.py:592-598), so the clip's scaling rides along in both -- the one-launch surfel
        Adam multiplies by the coefficient on the way in, torch's fused AdamW divides by its `grad_scale` = 1 / coefficient --
        and norm + coefficient + its inverse come from ONE launch over the flat surfel buffer and the networks' tensors,
        instead of clip_grad_norm_'s per-tensor norms, stack, norm, clamp and a multiply pass over 47 MB.  While the networks'
        gradients still ACCUMULATE over a round (before optim_warp_neus_iters) upstream's in-place clip scales what has
        accumulated, every step: that stays torch's clip_grad_norm_."""
        from ..gs.surfel_optim import NetworkAdamW, SurfelAdam
        return (self._adamw_steps(step) and isinstance(self.gs_optimizer, SurfelAdam) and self.world == 1
                and bool(self.opts.get("fold_clip_into_optimizers", True))
                and (isinstance(self.optimizer, NetworkAdamW) or any(g.get("fused") for g in self.optimizer.param_groups)))

    def clip_gradients(self, max_norm: float = 5.0, step: int | None = None):
        """clip_grad_norm_ over the parameters that have a gradient (trainer.py:861-869)."""
        if step is not None and self._fold_clip_into_both(step):
            from ..gs.surfel_optim import clip_coef
            if self._flat is not None and all(self._bound(p) for p in self.exchanged_params() if p.grad is not None):
                grads = [self._flat]        # (every surfel gradient is a view of it; the networks' places in it stay zero)
            else:
                grads = [p.grad for p in self.exchanged_params() if not any(p is q for q in self._net_params)]
            grads = grads + [p.grad for p in self._net_params]
            if not any(g is not None for g in grads):
                return None
            norm, self._clip_coef, self._clip_inv = clip_coef(grads, max_norm, with_inverse=True)
            return norm
        if self._fold_clip_into_adam():
            # norm and coefficient from one launch; the surfel Adam multiplies the gradients by the coefficient on the
            # way in (csrc/optim.hip)
            from ..gs.surfel_optim import clip_coef
            sq = self.__dict__.pop("_chunk_sq", None)
            scale = self.__dict__.pop("_mean_scale", 1.0)
            if sq:  # (after an exchange: the chunk norms are there, the gradients still hold the sum over the ranks)
                norm = torch.stack(sq).sum().sqrt() * scale
                self._clip_coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0) * scale
                return norm
            grads = [self._flat] if self._flat is not None else [p.grad for p in self.exchanged_params()]
            if not any(g is not None for g in grads):
                return None
            norm, self._clip_coef = clip_coef(grads, max_norm)
            return norm
        if self._flat is None or self.optim_warp or self._rest_slot is not None:
            return torch.nn.utils.clip_grad_norm_(self.exchanged_params(), max_norm)
        norm = torch.linalg.vector_norm(self._flat)
        self._flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm

    def _fold_clip_into_adam(self):
        """The clip's scaling pass and the next step's zero fill ride along in the one-launch Adam when nothing but that
        Adam reads the gradients after the clip: surfels on the GPU, frozen networks."""
        from ..gs.surfel_optim import SurfelAdam
        return isinstance(self.gs_optimizer, SurfelAdam) and self.optimizer is None

    def _optimizer_step(self, step: int):
        """The surfel Adam (and, once it is due, the networks' optimizer) on the clipped gradients."""
        coef = self.__dict__.pop("_clip_coef", None)
        if coef is not None:
            # every exchanged gradient is a live view of the flat buffer unless a parameter was re-created this step
            # (densify / prune / reset_opacity): only then is the buffer not left zero-filled
            whole = self._flat is not None and all(p.grad is not None for p in self.exchanged_params())
            self.gs_optimizer.step(grad_scale=coef, zero_grads=whole)
            self._flat_is_zero = whole
        else:
            self.gs_optimizer.step()
        if self.optimizer is not None and step >= self.optim_warp_from:
            from ..gs.surfel_optim import NetworkAdamW
            inv = self.__dict__.pop("_clip_inv", None)
            if isinstance(self.optimizer, NetworkAdamW):
                # (the clip, folded in: this kernel multiplies by the coefficient, as the surfel Adam does)
                self.optimizer.step(grad_scale=coef if inv is not None else None)
            else:
                if inv is not None:
                    self.optimizer.grad_scale = inv     # (torch's fused step divides the gradients by it)
                try:
                    self.optimizer.step()
                finally:
                    if inv is not None:
                        del self.optimizer.grad_scale
            self.scheduler.step()
            self._net_accum = [None] * len(self._net_params)   # optimizer.zero_grad() (trainer.py:596-598)

    def _sync_densification_stats(self):
        if self.world == 1:
            return
        m = self.model
        dist.all_reduce(m.xyz_gradient_accum, op=dist.ReduceOp.SUM)
        dist.all_reduce(m.denom, op=dist.ReduceOp.SUM)
        dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX)

    # ---- one optimizer step on this rank's frames
    def _forward_backward(self, batch: dict, step: int) -> dict:
        m = self.model
        # the depth / normal maps are only read by the regularisers, whose weights are 0 until step 8000
        need_geometry = step > 8000 and (self.cfg.lambda_normal != 0.0)
        M = int(batch["frameid"].shape[0])
        from .. import _lib as _native_lib
        if m._xyz.is_cuda and M <= _native_lib.LOSS_MAX_FRAMES and m.opts.get("fused_loss", True) and \
                (not need_geometry or m.opts.get("fused_normal_loss", True)):
            # colour / silhouette / distortion / normal-consistency terms and their gradient planes in five launches
            # (csrc/loss.hip); with the normal term on, the depth / normal post-processing of render() runs inside them
            from .loss_fused import stage3_loss
            # colour and silhouette terms read the colour and the alpha plane; the distortion term (plane 6) only counts
            # once lambda_dist does, the normal term reads planes 0-5: until then the blend kernels carry nothing else
            # (aux_planes, csrc/blend.hip BLEND_LITE).  With the normal term on but the upstream defaults lambda_dist = 0
            # (lab4d/config.py:181) and depth_ratio = 0 (gs/arguments/__init__.py:68; gs/gaussian_renderer/__init__.py:146:
            # the surface depth is the EXPECTED depth) only planes 0-4 -- depth, alpha, normal -- carry gradient
            # (lab4d/engine/model.py:817-842): the blend kernels then leave the median sample and the distortion moments out
            # (BLEND_GEOM) and, where they run segment-parallel, need no transmittance pre-pass.
            from ..diff_surfel_rasterization import AUX_ALPHA, AUX_GEOM
            lam_d = float(self.cfg.lambda_dist) if step > 8000 else 0.0
            depth_ratio = float(getattr(m.pipeline, "depth_ratio", 0.0))
            if lam_d == 0.0 and not need_geometry and m.opts.get("alpha_only_blend", True):
                aux = AUX_ALPHA
            elif lam_d == 0.0 and depth_ratio == 0.0 and m.opts.get("geom_blend", True):
                aux = AUX_GEOM
            else:
                aux = 0
            rendered = m.render_frames(batch["frameid"], batch["Kinv"], batch["H"], batch["W"], outputs=("raw",),
                                       aux_planes=aux)
            if "raw_stacked" in rendered:   # the frames came out of one stacked launch set: (3,M,H,W), (8,M,H,W)
                colors, allmaps = rendered["raw_stacked"]
            else:
                colors, allmaps = zip(*rendered["raw"])
            from .loss_fused import unit_gradient
            cams = m.get_gs_Kcamera(batch["Kinv"], batch["H"], batch["W"]) if need_geometry else None  # (cached)
            losses = stage3_loss(colors, allmaps, getattr(m, "learnable_bkgd", None), batch, step, self.cfg,
                                 cameras=cams, depth_ratio=float(getattr(m.pipeline, "depth_ratio", 0.0)))
            total = losses.pop("total")  # (summed by the kernel; the backward starts from a cached 1.0)
            with self._gradient_outputs():
                total.backward(gradient=unit_gradient(m._xyz.device))
            return losses
        else:
            outputs = None if need_geometry else ("render", "acc", "rend_dist")
            rendered = m.render_frames(batch["frameid"], batch["Kinv"], batch["H"], batch["W"], outputs=outputs)
            losses = compute_losses(rendered, batch, step, self.cfg)
        total = sum(losses.values())
        total.backward()
        return losses

    # ---- plain steps as one captured hipGraph (lab4d/captured_step.py)
    def _plain_step(self, step: int) -> bool:
        """No host decision inside the step: no SH-degree raise, no densify / prune / opacity reset / outlier pass, and -- with
        networks that train -- AdamW stepping (from optim_warp_neus_iters on) and no start of a round."""
        c, m = self.cfg, self.model
        if step % 1000 == 0 and m.active_sh_degree < m.max_sh_degree:
            return False
        if step < c.densify_until_iter:
            if step > c.densify_from_iter and not self._capture_while_densifying:
                return False          # ("auto": the densification regime stays eager, see __init__)
            if step > c.densify_from_iter and step % c.densification_interval == 0:
                return False
            if step % c.opacity_reset_interval == 0:
                return False
            if c.densify_from_iter < step < c.outlier_stop_iter and step % c.outlier_filtering_interval == 0:
                return False
        if self.optimizer is not None and (step < self.optim_warp_from or step % self.iters_per_round == 0):
            return False
        return True

    def _capture_key(self, batch: dict, step: int):
        m = self.model
        K = batch["Kinv"]
        cams = (tuple(float(x) for x in K.reshape(-1).tolist()) if not K.is_cuda else (K.data_ptr(), K._version),
                tuple(int(h) for h in batch["H"]), tuple(int(w) for w in batch["W"]))
        shapes = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor) and v.is_cuda)
        nets = tuple(p._version for p in m._warp_param_list()) if self.optimizer is None else ()
        # (every surfel tensor: densify / prune re-create all of them, reset_opacity the opacity alone)
        return (tuple(p.data_ptr() for p in self.surfel_params()), int(m._xyz.shape[0]), int(m.active_sh_degree), step > 8000,
                step < self.cfg.densify_until_iter, cams, shapes, nets)

    def _settle_captured(self):
        """The verdict of the captured step in flight (its forward finished long ago: the header copy sits in pinned memory):
        a step that did not fit its buffers changed nothing on the device -- take the host's books of it back and run it
        eagerly, as the eager loop's replay would have, before anything else happens."""
        fl, self._inflight = self._inflight, None
        if fl is None:
            return
        cs, batch = fl
        if cs.verdict():
            return
        self.captured_stats["taken_back"] += 1
        torch.cuda.current_stream(self.model._xyz.device).synchronize()
        cs.take_back()
        self.current_steps -= 1
        self._captured.clear()          # (the hints have moved: what was captured with the old ones goes)
        self._streak = (None, 0)
        self._train_step_eager(batch)

    def _step_stream(self):
        dev = self.model._xyz.device
        if self._cap_stream is None or self._cap_stream.device != dev:
            self._cap_stream = torch.cuda.Stream(dev)
        return self._cap_stream

    def train_step(self, batch: dict) -> dict:
        if not self.captured_step:
            return self._train_step_eager(batch)
        # Every step of a trainer whose plain steps are captured -- eager ones included -- runs on ONE side stream: a graph
        # is captured on a non-default stream, and autograd keeps a parameter's gradient-accumulation node on the stream it
        # was first used on; eager steps on the default stream would leave nodes there that a later capture trips over.
        s, cur = self._step_stream(), torch.cuda.current_stream(self.model._xyz.device)
        if cur == s:
            return self._train_step_on_stream(batch)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            out = self._train_step_on_stream(batch)
        cur.wait_stream(s)
        return out

    def settle(self):
        """The verdict of a captured step still in flight (train_step asks for it at the start of the NEXT step): call before
        reading the model after the last step of a run (checkpoints, evaluation)."""
        if not self.captured_step or self._inflight is None:
            return
        s, cur = self._step_stream(), torch.cuda.current_stream(self.model._xyz.device)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            self._settle_captured()
        cur.wait_stream(s)

    def _train_step_on_stream(self, batch: dict) -> dict:
        self._settle_captured()
        step = self.current_steps
        if not self._plain_step(step):
            self._streak = (None, 0)
            return self._train_step_eager(batch)
        key = self._capture_key(batch, step)
        cs = self._captured.get(key)
        if cs is None:
            last, n = self._streak
            n = n + 1 if last == key else 1
            self._streak = (key, n)
            if n <= self.capture_after:
                return self._train_step_eager(batch)
            from .captured_step import CapturedStep
            if len(self._captured) >= 4:
                self._captured.clear()
            import time as _time
            t0 = _time.perf_counter()
            cs = self._captured[key] = CapturedStep(self, batch, step)
            self.captured_stats["captures"] += 1
            self.captured_stats["capture_ms"] = self.captured_stats.get("capture_ms", 0.0) + 1e3 * (_time.perf_counter() - t0)
        losses = cs.replay(batch)
        self.captured_stats["replays"] += 1
        self._inflight = (cs, batch)
        self.current_steps += 1
        return losses

    def _train_step_eager(self, batch: dict) -> dict:
        m, c = self.model, self.cfg
        step = self.current_steps
        if step % 1000 == 0:
            m.oneupSHdegree()  # trainer.py:465-466 (also at step 0, as upstream); update_learning_rate (:464) is a
            #                    no-op upstream -- its `"._xyz" in param_group["params"]` test never holds
        if m._xyz.is_cuda:
            # The rasterizer's only host wait (the pair count that sizes the binning buffer) is deferred to
            # one check per step: if a frame outgrew its buffer -- it then rendered only the background --
            # the gradients of this step are dropped and the step is replayed with exact buffers.
            from .. import _C
            # (everything below runs under the MODEL's rasterizer context: its hints, its unchecked forwards, the gradient
            # outputs bound for its backward -- another model's step, or another thread's, has its own)
            with m.raster_context:
                self.begin_gradients()
                with _C.deferred_capacity_check():
                    losses = self._forward_backward(batch, step)
                if not _C.check_deferred():
                    self.begin_gradients()
                    losses = self._forward_backward(batch, step)
        else:
            self.begin_gradients()
            losses = self._forward_backward(batch, step)
        if self.world == 1:
            self._fold_net_gradients(from_slots=False, adopt=self.optimizer is not None and step >= self.optim_warp_from)
        self.allreduce_gradients(async_op=True)   # in flight while the statistics below are gathered
        self.gather_densification_stats(step)
        self.finish_step(step)
        return {k: v.detach() for k, v in losses.items()}

    def gather_densification_stats(self, step: int, keep=None):
        """max_radii2D / xyz_gradient_accum / denom from the frames rendered last (trainer.py:549-556); does not read the
        parameter gradients, so it runs while the exchange is in flight.  keep (captured steps): a device bool -- False when the
        step's forward did not fit its buffers, the statistics of such a step are not taken (the eager loop replays it first)."""
        m = self.model
        if step >= self.cfg.densify_until_iter:
            return
        with torch.no_grad():
            for i in range(len(m._radii_batch)):
                vis, radii = m._visibility_filter_batch[i], m._radii_batch[i]
                if keep is not None:
                    vis = vis & keep
                m.max_radii2D.copy_(torch.where(vis, torch.maximum(m.max_radii2D, radii.float()), m.max_radii2D))
                m.add_densification_stats(m._viewspace_points_batch[i], vis)

    def finish_step(self, step: int):
        """Everything after the step's gradients exist: join the exchange, clip, the densify / prune / opacity-reset
        cadence, the optimizers (trainer.py:547-598)."""
        m, c = self.model, self.cfg
        with torch.no_grad():
            self.wait_gradients()
            self.clip_gradients(5.0, step)        # check_grad precedes the densification upstream (trainer.py:547)
            if step < c.densify_until_iter:
                gen = None
                if step > c.densify_from_iter and step % c.densification_interval == 0:
                    self._sync_densification_stats()
                    gen = torch.Generator(device=m._xyz.device).manual_seed(1000003 * step + 17)
                    size_threshold = 20 if step > c.opacity_reset_interval else None
                    if m._xyz.is_cuda:  # decisions, row gather and moment surgery on the device (csrc/optim.hip)
                        from ..gs.surfel_optim import densify_and_prune_fused
                        densify = lambda *a, **k: densify_and_prune_fused(m, *a, **k)  # noqa: E731
                    else:
                        densify = m.densify_and_prune
                    densify(c.densify_grad_threshold, 0.005, m.cameras_extent, size_threshold, generator=gen)
                    if step % (10 * c.densification_interval) == 0:
                        densify(c.densify_grad_threshold * 0.1, 0.002, m.cameras_extent * 100, size_threshold,
                                generator=gen)
                if step % c.opacity_reset_interval == 0:  # (step 0 included, as upstream: trainer.py:570)
                    m.reset_opacity()
                if (c.densify_from_iter < step < c.outlier_stop_iter and step % c.outlier_filtering_interval == 0
                        and (m._xyz.is_cuda or self.outlier_neighbor_count is not None)):
                    # trainer.py:573-588: open3d remove_radius_outlier(nb_points=20, radius=0.004) on the CPU
                    # upstream; here the neighbour count is a HIP kernel and nothing leaves the GPU.  Every rank
                    # holds the same surfels, so every rank prunes the same ones.  (`outlier_neighbor_count`: a stand-in
                    # for surfels that live on the CPU -- the gloo tests plant scipy's cKDTree; the product has none.)
                    count = self.outlier_neighbor_count
                    if count is None:
                        from ..simple_knn import radius_neighbor_count as count
                    m.prune_points(count(m.get_xyz, self.outlier_radius) <= self.outlier_nb_points)
        # (parameters re-created by densify / prune / reset_opacity have no gradient and are skipped, as upstream)
        self._optimizer_step(step)
        for p in self.surfel_params():
            p.grad = None
        self.current_steps += 1


def make_intrinsics_inv(M: int, H: int, W: int, tanfov: float = 0.5, device="cpu") -> torch.Tensor:
    """Kinv of a centred pinhole camera (--force_center_cam, model.py:420-425): maps pixel (u,v,1) to
    the ray (x/z, y/z, 1)."""
    fx, fy = W / (2 * tanfov), H / (2 * tanfov * H / W)
    K = torch.tensor([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], device=device)
    return torch.inverse(K)[None].expand(M, -1, -1).contiguous()


def synthetic_batch(model: DeformableSurfels, frame_ids, H: int, W: int, seed: int = 0) -> dict:
    """A frame batch with random target images (throughput runs; no dataset ships with the reference)."""
    dev = model._xyz.device
    M = len(frame_ids)
    g = torch.Generator().manual_seed(seed)
    # (Kinv stays on the host: cameras are built there, deformable_surfels.get_gs_Kcamera)
    fid = torch.as_tensor(frame_ids, device=dev)
    if M and not isinstance(frame_ids, torch.Tensor):
        fid._vidu4d_host_range = (int(min(frame_ids)), int(max(frame_ids)))   # (DeformableSurfels._check_frame_ids)
    return {"frameid": fid, "Kinv": make_intrinsics_inv(M, H, W, device="cpu"),
            "H": [H] * M, "W": [W] * M, "rgb": torch.rand(M, H, W, 3, generator=g).to(dev),
            "mask": (torch.rand(M, H, W, 1, generator=g) > 0.5).float().to(dev),
            "vis2d": torch.ones(M, H, W, 1, device=dev)}
