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
    def __init__(self, model: DeformableSurfels, opts: dict | None = None):
        self.model = model
        self.cfg = _Args(opts or model.opts)
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
        # --gs_optim_warp=False (the README's Stage-3 command): warp and camera networks come from the
        # Stage-2 checkpoint and are never stepped (trainer.py:592-598).  Upstream still back-propagates
        # into them; freezing them is results-equivalent for the surfels up to the gradient clip (upstream's
        # check_grad, :861-869, puts the unused warp gradients into the norm it clips by -- DESIGN.md §5) and skips
        # the weight-gradient GEMMs and the (M,N,B,.) broadcast reductions of the warp backward.
        o = opts or model.opts
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
            groups, lrs = [], []
            for n, prm in net_params:
                groups.append({"params": [prm], "name": n})
                lrs.append(c.learning_rate * (10.0 if any(n.endswith(e[1:]) or e in n for e in explicit) else 1.0))
            total = max(2, int(o.get("num_rounds", 1)) * int(o.get("iters_per_round", 200)))
            self.optimizer = torch.optim.AdamW(groups, lr=c.learning_rate, betas=(0.9, 0.999), weight_decay=1e-4)
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
                self.optimizer, lrs, total, pct_start=min(0.5, 2.0 / max(1, int(o.get("num_rounds", 1)))),
                cycle_momentum=False, anneal_strategy="linear", div_factor=25.0, final_div_factor=1.0)
        self._net_params = [prm for _, prm in net_params]

    # ---- the path's only exchange
    def surfel_params(self):
        m = self.model
        ps = [m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m._regist_feat]
        if self.cfg.gs_learnable_bg:
            ps.append(m.learnable_bkgd)
        return ps

    def exchanged_params(self):
        """Everything the ranks must agree on after a step: the surfels, and the networks when they train."""
        return self.surfel_params() + (self._net_params if self.optim_warp else [])

    def bind_flat_gradients(self):
        """Makes every exchanged parameter's .grad a view of ONE persistent fp32 buffer and zeroes it (this is
        the step's zero_grad): autograd then accumulates straight into the buffer the all-reduce, the norm for
        the clip and the fused Adam read -- no gather / scatter copies around the collective.  Re-bound every
        step because densify / prune re-create the surfel parameters."""
        ps = self.exchanged_params()
        n = sum(p.numel() for p in ps)
        if self._flat is None or self._flat.numel() != n or self._flat.device != ps[0].device:
            self._flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        elif not self.__dict__.pop("_flat_is_zero", False):  # (the one-launch Adam left it zero-filled)
            self._flat.zero_()
        off = 0
        for p in ps:
            p.grad = self._flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        return self._flat

    def _flat_needed(self):
        """One rank, surfels on the GPU, frozen networks: nothing reads the gradients but the one-launch clip and the
        one-launch Adam, which take per-tensor pointers -- autograd then hands every parameter its gradient tensor as it
        is (no accumulation into a pre-bound buffer: 16 read-add-write passes per step) and there is nothing to zero."""
        return not (self.world == 1 and self._fold_clip_into_adam())

    def begin_gradients(self):
        """The step's zero_grad: the flat buffer where it is needed (bind_flat_gradients), else plain `grad = None`."""
        if self._flat_needed():
            return self.bind_flat_gradients()
        self._flat = None
        for p in self.exchanged_params():
            p.grad = None
        return None

    def allreduce_gradients(self, async_op: bool = False):
        """Sum of the flat gradient buffer over the ranks (the mean is folded into clip_gradients).  With async_op
        the collective is left in flight (RCCL runs it on its own stream) and `wait_gradients` joins it, so that
        work that does not read the gradients -- the densification statistics -- overlaps it."""
        if self.world == 1:
            return
        if self._flat is None or any(p.grad is None or p.grad.untyped_storage().data_ptr() != self._flat.untyped_storage().data_ptr()
                                     for p in self.exchanged_params()):
            # gradients that were not produced into the flat buffer (a caller that set them by hand): gather them
            ps = self.exchanged_params()
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
            self.bind_flat_gradients()
            for p, g_ in zip(ps, grads):
                p.grad.copy_(g_)
        self._pending_reduce = dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, async_op=True)
        if not async_op:
            self.wait_gradients()

    def wait_gradients(self):
        if self._pending_reduce is not None:
            self._pending_reduce.wait()
            self._pending_reduce = None
            self._flat.div_(self.world)

    def clip_gradients(self, max_norm: float = 5.0):
        """clip_grad_norm_ over the exchanged parameters (trainer.py:861-869), one norm over the flat buffer."""
        if self._fold_clip_into_adam():
            # norm and coefficient from one launch; the surfel Adam multiplies the gradients by the coefficient on the
            # way in (csrc/optim.hip)
            from ..gs.surfel_optim import clip_coef
            grads = [self._flat] if self._flat is not None else [p.grad for p in self.exchanged_params()]
            if not any(g is not None for g in grads):
                return None
            norm, self._clip_coef = clip_coef(grads, max_norm)
            return norm
        if self._flat is None:
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
            self.optimizer.step()
            self.scheduler.step()

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
        if m._xyz.is_cuda and not need_geometry and M <= 8 and m.opts.get("fused_loss", True):
            # colour / silhouette / distortion terms and their gradient planes in five launches (csrc/loss.hip)
            from .loss_fused import stage3_loss
            # colour and silhouette terms read the colour and the alpha plane; the distortion term (plane 6) only counts
            # once lambda_dist does: until then the blend kernels carry nothing else (aux_planes, csrc/blend.hip LITE)
            from ..diff_surfel_rasterization import AUX_ALPHA
            lam_d = float(self.cfg.lambda_dist) if step > 8000 else 0.0
            aux = AUX_ALPHA if (lam_d == 0.0 and m.opts.get("alpha_only_blend", True)) else 0
            rendered = m.render_frames(batch["frameid"], batch["Kinv"], batch["H"], batch["W"], outputs=("raw",),
                                       aux_planes=aux)
            if "raw_stacked" in rendered:   # the frames came out of one stacked launch set: (3,M,H,W), (8,M,H,W)
                colors, allmaps = rendered["raw_stacked"]
            else:
                colors, allmaps = zip(*rendered["raw"])
            from .loss_fused import unit_gradient
            losses = stage3_loss(colors, allmaps, getattr(m, "learnable_bkgd", None), batch, step, self.cfg)
            total = losses.pop("total")  # (summed by the kernel; the backward starts from a cached 1.0)
            z = self.__dict__.get("_zero_loss")
            if z is None or z.device != m._xyz.device:
                z = self._zero_loss = torch.zeros((), device=m._xyz.device)
            losses["normal_loss"] = z
            total.backward(gradient=unit_gradient(m._xyz.device))
            return losses
        else:
            outputs = None if need_geometry else ("render", "acc", "rend_dist")
            rendered = m.render_frames(batch["frameid"], batch["Kinv"], batch["H"], batch["W"], outputs=outputs)
            losses = compute_losses(rendered, batch, step, self.cfg)
        total = sum(losses.values())
        total.backward()
        return losses

    def train_step(self, batch: dict) -> dict:
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
            self.begin_gradients()
            with _C.deferred_capacity_check():
                losses = self._forward_backward(batch, step)
            if not _C.check_deferred():
                self.begin_gradients()
                losses = self._forward_backward(batch, step)
        else:
            self.begin_gradients()
            losses = self._forward_backward(batch, step)
        self.allreduce_gradients(async_op=True)   # in flight while the statistics below are gathered

        with torch.no_grad():
            if step < c.densify_until_iter:
                for i in range(len(m._radii_batch)):
                    vis, radii = m._visibility_filter_batch[i], m._radii_batch[i]
                    m.max_radii2D.copy_(torch.where(vis, torch.maximum(m.max_radii2D, radii.float()), m.max_radii2D))
                    m.add_densification_stats(m._viewspace_points_batch[i], vis)
            self.wait_gradients()
            self.clip_gradients(5.0)              # check_grad precedes the densification upstream (trainer.py:547)
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
                if (m._xyz.is_cuda and c.densify_from_iter < step < c.outlier_stop_iter
                        and step % c.outlier_filtering_interval == 0):
                    # trainer.py:573-588: open3d remove_radius_outlier(nb_points=20, radius=0.004) on the CPU
                    # upstream; here the neighbour count is a HIP kernel and nothing leaves the GPU.  Every rank
                    # holds the same surfels, so every rank prunes the same ones.
                    from ..simple_knn import radius_neighbor_count
                    m.prune_points(radius_neighbor_count(m.get_xyz, 0.004) <= 20)
        # (parameters re-created by densify / prune / reset_opacity have no gradient and are skipped, as upstream)
        self._optimizer_step(step)
        for p in self.exchanged_params():
            p.grad = None
        self.current_steps += 1
        return {k: v.detach() for k, v in losses.items()}


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
    return {"frameid": torch.as_tensor(frame_ids, device=dev), "Kinv": make_intrinsics_inv(M, H, W, device="cpu"),
            "H": [H] * M, "W": [W] * M, "rgb": torch.rand(M, H, W, 3, generator=g).to(dev),
            "mask": (torch.rand(M, H, W, 1, generator=g) > 0.5).float().to(dev),
            "vis2d": torch.ones(M, H, W, 1, device=dev)}
