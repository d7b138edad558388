"""A plain Stage-3 fitting step as ONE captured hipGraph: warp -> stacked rasterizer -> loss kernels ->
backward -> densification statistics -> clip -> surfel Adam (+ the networks' AdamW when they train) replayed with one
`hipGraphLaunch` instead of ~60 (frozen networks) / ~250 (training networks) launches driven by ~0.8 / ~2.3 ms of Python.

Reference loop: /root/reference/lab4d/engine/trainer.py:439-602 (train_one_round), one `update_aux_vars -> forward ->
backward -> check_grad -> densify cadence -> optimizer.step` per iteration.  What a captured step must not lose against the
eager one (Stage3Trainer.train_step), and how each point is kept:

* WHICH steps: only "plain" ones -- no SH-degree raise, no densify / prune / opacity reset / outlier pass, no start of a
  round with training networks, one rank.  Everything else runs eagerly as before; the graph is keyed on what its launches
  bake (surfel tensors, SH degree, regime of the regularisers, frame count, image size, cameras, which optimizers step) and
  re-captured when the key changes (a densify step re-creates the surfel tensors: new key).
* The rasterizer's ONE host decision -- did the frame fit its binning buffer / segment limit (else it rendered the background
  only and the step must be run again with exact buffers, _C.deferred_capacity_check) -- cannot be asked mid-graph.  The
  graph derives the verdict on the DEVICE from the header words the forward leaves (pair count against the captured capacity,
  `truncated`) into a skip word; every launch that changes persistent state reads it: the surfel Adam
  (vidu4d_adam_step_guarded), the networks' fused AdamW (its `found_inf` input, which also takes its step counter back), the
  densification statistics (masked).  A skipped step changes NOTHING; the host learns of it one step later from the header
  copy the forward queues into pinned memory (long there by then: no wait), takes its own bookkeeping of that step back and
  runs it eagerly -- so the trajectory is the eager loop's, in the same order.
* Per-step scalars that eager launches pass by value: the Adam bias corrections come from one small async copy in front of the
  replay (gs/surfel_optim.CapturedScalars: the same float values), the networks' learning rates are device tensors the
  scheduler fills (torch's capturable AdamW).
* The step's inputs (frame ids, target images, masks) are copied into static tensors in front of the replay.
"""
from __future__ import annotations

import time

import torch

from .. import _C
from ..gs.surfel_optim import CapturedScalars

SENTINEL = -1   # what the pinned header slots hold between "replay queued" and "the forward's header copy has landed"


class CapturedStep:
    def __init__(self, trainer, batch: dict, step: int):
        self.tr = tr = trainer
        m = tr.model
        dev = m._xyz.device
        self.static = {k: torch.empty_like(v) for k, v in batch.items() if isinstance(v, torch.Tensor) and v.is_cuda}
        self.host_items = {k: v for k, v in batch.items() if k not in self.static}   # (Kinv, H, W: part of the key)
        self.skip = torch.zeros(1, dtype=torch.int32, device=dev)         # non-zero: this replay changes nothing
        self.skip_f = torch.zeros((), dtype=torch.float32, device=dev)    # ... as the fused AdamW's `found_inf` (0-dim, as its step counters)
        self.scalars = CapturedScalars(dev, skip=self.skip)
        self.adamw = tr.optimizer is not None and step >= tr.optim_warp_from
        from ..gs.surfel_optim import NetworkAdamW
        # (the networks' AdamW as the surfel optimizer's kernel: its scalars and skip word like the surfel Adam's)
        self.net_scalars = CapturedScalars(dev, skip=self.skip) if self.adamw and isinstance(tr.optimizer, NetworkAdamW) else None
        self.replays = 0
        s = tr._step_stream()
        # what the step's launches create lazily PER STREAM must exist before the capture (made inside it, it would live in
        # the graph's private pool -- or be a host allocation in the middle of a capture): the clip kernel's counters, the
        # pinned slots the forwards' headers are copied to
        from ..gs.surfel_optim import ensure_clip_workspace
        ensure_clip_workspace(dev, s.cuda_stream)
        rc = m.raster_context
        M = int(batch["frameid"].shape[0])
        for k in range((M + 7) // 8):
            rc.pinned.setdefault((str(dev), s.cuda_stream, k), torch.zeros(16, dtype=torch.int32).pin_memory())
        self._load(batch)
        self.graph = torch.cuda.CUDAGraph()
        inline, m.opts["graphed_warp_networks"] = m.opts.get("graphed_warp_networks", True), "inline"
        try:
            with torch.cuda.graph(self.graph, stream=s):
                self.losses = self._body(step)
        finally:
            m.opts["graphed_warp_networks"] = inline
        # (what the capture left on the host's side: the parameters' .grad are the graph's own tensors now; nothing has run)

    def _load(self, batch):
        fid = batch["frameid"]
        rng = getattr(fid, "_vidu4d_host_range", None)
        for k, t in self.static.items():
            t.copy_(batch[k], non_blocking=True)
        if rng is None:   # (ids of unknown range: one device read -- producers of frame batches note theirs on the host)
            rng = (int(fid.min()), int(fid.max()))
        sf = self.static["frameid"]
        sf._vidu4d_host_range = rng
        if hasattr(sf, "_vidu4d_rows_checked"):
            del sf._vidu4d_rows_checked

    def _batch(self):
        return dict(self.host_items, **self.static)

    def _body(self, step):
        """The plain step, as Stage3Trainer.train_step + finish_step run it, minus the host's bookkeeping (step counts, the
        scheduler, current_steps: `replay` does those per replay)."""
        tr = self.tr
        m = tr.model
        with m.raster_context:
            # (the flat gradient buffer, where there is one: left zero-filled by the Adam launch of the step before -- eager or
            # replayed -- as in the eager loop; a skipped replay leaves garbage there and `take_back` says so)
            tr.begin_gradients()
            with _C.graph_capture_mode() as cap:
                losses = tr._forward_backward(self._batch(), step)
        self.frames = cap.frames
        bad = None
        for hdr, capacity in cap.headers:
            b = (hdr[0] > int(capacity)) | (hdr[6] != 0)
            bad = b if bad is None else (bad | b)
        self.skip.copy_(bad.to(torch.int32).reshape(1))
        self.skip_f.copy_(bad.to(torch.float32).reshape(()))
        tr._fold_net_gradients(from_slots=False, adopt=self.adamw)
        tr.gather_densification_stats(step, keep=~bad)
        with torch.no_grad():
            tr.clip_gradients(5.0, step)
            coef = tr.__dict__.pop("_clip_coef", None)
            inv = tr.__dict__.pop("_clip_inv", None)
            whole = tr._flat is not None and all(p.grad is not None for p in tr.exchanged_params())
            tr.gs_optimizer.step(grad_scale=coef, zero_grads=whole, captured=self.scalars)
            self.leaves_flat_zero = whole
            if self.adamw and self.net_scalars is not None:
                tr.optimizer.step(grad_scale=coef if inv is not None else None, captured=self.net_scalars)
            elif self.adamw:
                tr.optimizer.found_inf = self.skip_f    # (read by torch's fused step: nothing is updated, its step counter taken back)
                if inv is not None:
                    tr.optimizer.grad_scale = inv       # (the clip, folded in: Stage3Trainer._fold_clip_into_both)
                try:
                    tr.optimizer.step()
                finally:
                    del tr.optimizer.found_inf
                    if inv is not None:
                        del tr.optimizer.grad_scale
        return {k: v.detach() for k, v in losses.items()}

    def replay(self, batch):
        tr = self.tr
        self._load(batch)
        m = tr.model
        m._check_frame_ids(self.static["frameid"], int(m.frame_offset_raw[-1]))
        for slot, _stat, _cap, _key in self.frames:
            slot[0] = SENTINEL
        self.scalars.advance()
        if self.net_scalars is not None:
            self.net_scalars.advance()
        self.graph.replay()
        if self.adamw:
            tr.scheduler.step()
            tr._net_accum = [None] * len(tr._net_params)
        if self.leaves_flat_zero:
            tr._flat_is_zero = True
        self.replays += 1
        return self.losses

    def verdict(self, timeout_s: float = 10.0) -> bool:
        """True when the last replay's forwards fit their buffers (waits -- normally not at all -- for their header copies)."""
        t0 = None
        for slot, _stat, _cap, _key in self.frames:
            while int(slot[0]) == SENTINEL:
                if t0 is None:
                    t0 = time.perf_counter()
                elif time.perf_counter() - t0 > timeout_s:   # (should never happen: the copy is a node of the graph)
                    self.tr._cap_stream.synchronize()
                    break
        return _C.check_slots(self.frames, context=self.tr.model.raster_context)

    def take_back(self):
        """The last replay was skipped on the device: the host's books of it."""
        tr = self.tr
        tr.__dict__.pop("_flat_is_zero", None)   # (the skipped Adam did not zero the flat gradient buffer)
        self.scalars.rewind(1)
        if self.net_scalars is not None:
            self.net_scalars.rewind(1)
        if self.adamw:
            tr.scheduler.last_epoch -= 2
            tr.scheduler.step()
