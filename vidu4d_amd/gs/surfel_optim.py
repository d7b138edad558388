"""The surfel optimizer and the densification step on the HIP path (csrc/optim.hip).

`SurfelAdam` IS a torch.optim.Adam -- same constructor, param_groups, per-parameter state {"step", "exp_avg",
"exp_avg_sq"} and state_dict, so GaussianModel's optimizer surgery and the checkpoint format are untouched -- whose
step() updates every group with one launch instead of one multi-tensor launch per group (the reference builds one
group per surfel attribute because each has its own learning rate: lab4d/engine/trainer.py:240-255).

`densify_and_prune_fused` produces what GaussianModel.densify_and_prune (gs/scene/gaussian_model.py:434-448 with
:270-356, :384-432 under it) produces -- same rows in the same order, same optimizer-state handling -- from three
launches and one host read of the new surfel count."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from .. import _lib


class SurfelAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False, foreach=False, fused=False)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, zero_grads=False, captured=None):
        """grad_scale: device scalar the gradients are multiplied by on the way in (the clip coefficient, instead of a
        pass of its own over the gradients); zero_grads: leave the gradient arrays zero-filled.
        captured (lab4d/captured_step.py): a `CapturedScalars` -- the launch is being captured into a hipGraph: the per-step
        scalars (learning rate, bias corrections) are read from its device rows instead of being passed by value, the launch
        carries its skip word, and the step counts are NOT advanced here (`captured.advance()` does that before every replay)."""
        if closure is not None:
            raise RuntimeError("SurfelAdam: closures are not supported")
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue  # (freshly re-created parameters are skipped, as torch does)
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("SurfelAdam: contiguous fp32 HIP parameters required")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if zero_grads and not p.grad.is_contiguous():
                    raise RuntimeError("SurfelAdam: zero_grads needs contiguous gradients")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if captured is not None:
                    rec = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          p.numel(), 0.0, 0.0, 0.0, captured.row(group, p, st))
                else:
                    st["step"] += 1
                    t = float(st["step"])
                    rec = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          p.numel(), float(group["lr"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), None)
                batches.setdefault((p.device, b1, b2, group["eps"]), []).append((rec, g))
        lib = _lib.load()
        skip = None if captured is None or captured.skip is None else captured.skip.data_ptr()
        for (dev, b1, b2, eps), items in batches.items():
            stream = torch.cuda.current_stream(dev).cuda_stream
            for i in range(0, len(items), _lib.ADAM_MAX_TENSORS):
                chunk = items[i:i + _lib.ADAM_MAX_TENSORS]
                arr = (_lib.AdamTensor * len(chunk))(*[c[0] for c in chunk])
                _lib.check(lib.vidu4d_adam_step_guarded(len(chunk), arr, b1, b2, eps,
                                                        None if grad_scale is None else grad_scale.data_ptr(),
                                                        int(bool(zero_grads)), skip, stream), "adam step")
        return None


class NetworkAdamW(torch.optim.AdamW):
    """torch.optim.AdamW for the warp / camera networks (reference: lab4d/engine/trainer.py:177-286) -- same constructor,
    param_groups, per-parameter state {"step", "exp_avg", "exp_avg_sq"}, so the one-cycle scheduler and the checkpoint
    format are untouched -- whose step() is the surfel optimizer's kernel with the decoupled weight decay in front
    (csrc/optim.hip adam_kernel, vidu4d_adamw_step_guarded): 32 tensors per launch, three launches for the bob networks'
    66 tensors where torch's fused capturable form takes seven of ~24 us each (145 us of a 2.6 ms step).  `step` is kept
    as a Python float in the state (a tensor found there -- a checkpoint written by torch's AdamW -- is converted on
    first use).  grad_scale / zero_grads / captured: as SurfelAdam.step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, zero_grads=False, captured=None):
        if closure is not None:
            raise RuntimeError("NetworkAdamW: closures are not supported")
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr = group["lr"]
            for p in group["params"]:
                if p.grad is None:
                    continue   # (a parameter autograd never touched is skipped, weight decay included: as torch does)
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("NetworkAdamW: contiguous fp32 HIP parameters required")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0.0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif isinstance(st["step"], torch.Tensor):
                    st["step"] = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if captured is not None:
                    rec = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          p.numel(), 0.0, 0.0, 0.0, captured.row(group, p, st))
                else:
                    st["step"] += 1
                    t = st["step"]
                    rec = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          p.numel(), float(lr), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t), None)
                batches.setdefault((p.device, b1, b2, group["eps"], group["weight_decay"]), []).append(rec)
        lib = _lib.load()
        skip = None if captured is None or captured.skip is None else captured.skip.data_ptr()
        for (dev, b1, b2, eps, wd), items in batches.items():
            stream = torch.cuda.current_stream(dev).cuda_stream
            for i in range(0, len(items), _lib.ADAMW_MAX_TENSORS):
                chunk = items[i:i + _lib.ADAMW_MAX_TENSORS]
                arr = (_lib.AdamTensor * len(chunk))(*chunk)
                _lib.check(lib.vidu4d_adamw_step_guarded(len(chunk), arr, b1, b2, eps, wd,
                                                         None if grad_scale is None else grad_scale.data_ptr(),
                                                         int(bool(zero_grads)), skip, stream), "adamw step")
        return None


class CapturedScalars:
    """The per-step scalars of a SurfelAdam launch that lives in a captured hipGraph: one device row {lr, 1 - beta1^t,
    sqrt(1 - beta2^t)} per parameter tensor, refreshed from the host's step counts by ONE small async copy in front of every
    replay -- the same float values `step()` passes by value when it runs eagerly, so a captured step updates bit for bit
    what the eager step updates.  `skip`: the device word the launch reads (non-zero: change nothing)."""
    MAX_ROWS = 16

    def __init__(self, device, skip=None):
        self.dev = torch.zeros(self.MAX_ROWS, 4, dtype=torch.float32, device=device)
        self.host = torch.zeros(2, self.MAX_ROWS, 4, dtype=torch.float32).pin_memory()   # (two: the copy queued for the last
        self._flip = 0                                                                   # replay may not have run yet)
        self.rows: list = []   # (group, [state dicts]) per row, in the order the capture met them
        self._row_of: dict = {}
        self.skip = skip

    def row(self, group, p, st) -> int:
        # (the tensors of one group at one step count share a row: the networks' 66 tensors are two rows)
        key = (id(group), float(st["step"]))
        i = self._row_of.get(key)
        if i is None:
            if len(self.rows) >= self.MAX_ROWS:
                raise RuntimeError("CapturedScalars: more (group, step count) pairs than rows")
            i = self._row_of[key] = len(self.rows)
            self.rows.append((group, []))
        self.rows[i][1].append(st)
        return self.dev[i].data_ptr()

    def advance(self):
        """One optimizer step on the host's side of the books: every row's step count + 1, its scalars to the device (queued
        on the current stream, in front of the replay that reads them)."""
        self._flip ^= 1
        h = self.host[self._flip]
        for i, (group, states) in enumerate(self.rows):
            for st in states:
                st["step"] += 1
            t = float(states[0]["step"])
            b1, b2 = group["betas"]
            h[i, 0], h[i, 1], h[i, 2] = float(group["lr"]), 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)
        self.dev.copy_(h, non_blocking=True)

    def rewind(self, steps: int = 1):
        """The last `steps` replays changed nothing (their skip word was set): take their step counts back."""
        for _, states in self.rows:
            for st in states:
                st["step"] -= steps


_clip_ws: dict = {}


def ensure_clip_workspace(dev, stream_id):
    """The clip kernel's arrival counters for (device, stream), zero-filled ONCE -- a capture on that stream must find them
    (a workspace first made inside a capture would live in the graph's private pool and die with it)."""
    key = (dev, stream_id)
    if key not in _clip_ws:
        _clip_ws[key] = torch.zeros(_lib.CLIP_WORKSPACE_FLOATS, dtype=torch.float32, device=dev)
    return _clip_ws[key]


def clip_coef(grads, max_norm: float, with_inverse: bool = False):
    """(norm, coef) device scalars of torch.nn.utils.clip_grad_norm_ over `grads` (contiguous fp32 HIP tensors) from ONE
    launch (csrc/optim.hip::clip_kernel): norm = the 2-norm over all of them, coef = min(1, max_norm / (norm + 1e-6)) --
    what SurfelAdam.step takes as grad_scale.  (lab4d/engine/trainer.py:861-869.)
    with_inverse: (norm, coef, 1 / coef) -- 1 / coef is the `grad_scale` of torch's fused AdamW, which divides by it."""
    grads = [g for g in grads if g is not None and g.numel()]
    if not grads:
        raise RuntimeError("clip_coef: no gradients")
    dev = grads[0].device
    if any((not g.is_cuda) or g.dtype != torch.float32 or g.device != dev for g in grads):
        raise RuntimeError("clip_coef: fp32 HIP tensors on one device required")
    grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = ensure_clip_workspace(*key)   # (zero-filled once; the kernel leaves the arrival counter zero)
    out = torch.empty(3, dtype=torch.float32, device=dev)
    lib = _lib.load()
    sq = None
    for i in range(0, len(grads), _lib.CLIP_MAX_TENSORS):
        chunk = grads[i:i + _lib.CLIP_MAX_TENSORS]
        if len(grads) > _lib.CLIP_MAX_TENSORS:  # (more tensors than one launch takes: combine the chunk norms)
            out = torch.empty(3, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * len(chunk))(*[g.data_ptr() for g in chunk])
        nums = (C.c_int64 * len(chunk))(*[g.numel() for g in chunk])
        _lib.check(lib.vidu4d_grad_clip_coef(len(chunk), ptrs, nums, float(max_norm), ws.data_ptr(),
                                             out.data_ptr(), key[1]), "grad clip")
        if len(grads) > _lib.CLIP_MAX_TENSORS:
            sq = out[0] * out[0] if sq is None else sq + out[0] * out[0]
    if sq is not None:
        norm = sq.sqrt()
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        return (norm, coef, 1.0 / coef) if with_inverse else (norm, coef)
    return (out[0], out[1], out[2]) if with_inverse else (out[0], out[1])


_ATTRS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "regist_feat")


def densify_and_prune_fused(gm, max_grad, min_opacity, extent, max_screen_size, generator=None, samples=None):
    """In-place equivalent of gm.densify_and_prune(...) for a GaussianModel on a HIP device.  `samples`
    (tests): the (2 * n_selected, 3) scaled draws the Python path would be given."""
    dev = gm._xyz.device
    if not gm._xyz.is_cuda:
        raise RuntimeError("densify_and_prune_fused: HIP tensors required")
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    N = gm._xyz.shape[0]
    groups = {g["name"]: g for g in gm.optimizer.param_groups if g["name"] in _ATTRS}
    names = [n for n in _ATTRS if n in groups]
    if N == 0:
        return
    counts = torch.empty(3, N, dtype=torch.int32, device=dev)
    accum, denom = gm.xyz_gradient_accum.contiguous(), gm.denom.contiguous()
    _lib.check(lib.vidu4d_densify_plan(N, accum.data_ptr(), denom.data_ptr(), gm._scaling.data_ptr(),
                                       gm._opacity.data_ptr(), float(max_grad), float(gm.percent_dense * extent),
                                       float(min_opacity), float(0.1 * extent) if max_screen_size else -1.0,
                                       counts.data_ptr(), stream), "densify plan")
    inc = torch.cumsum(counts, dim=1, dtype=torch.int32)
    n_orig, n_clone, n_split = (int(v) for v in inc[:, -1].tolist())  # the one host read of this step
    rows = n_orig + n_clone + 2 * n_split
    if samples is not None:
        # the Python path's layout: row k / n_sel + k = first / second copy of the k-th split-selected surfel
        # (selected BEFORE the final prune); re-indexed by source surfel here
        g = accum / denom
        g[g.isnan()] = 0.0
        sel = (g.squeeze(-1) >= max_grad) & (gm.get_scaling.max(dim=1).values > gm.percent_dense * extent)
        draws = torch.zeros(2, N, 3, device=dev)
        draws[:, sel] = samples.to(dev).view(2, -1, 3)
        scaled = 1
    else:
        draws = torch.randn(2, N, 3, device=dev, generator=generator)
        scaled = 0
    src_row = torch.empty(max(rows, 1), dtype=torch.int32, device=dev)
    kind = torch.empty(max(rows, 1), dtype=torch.uint8, device=dev)
    recs, new_params, new_states, keep_alive = [], {}, {}, []
    for name in names:
        old = groups[name]["params"][0]
        width = old[0].numel() if N else 1
        src = old.detach().contiguous()
        dst = torch.empty((rows,) + tuple(old.shape[1:]), dtype=torch.float32, device=dev)
        st = gm.optimizer.state.get(old)
        has = st is not None and "exp_avg" in st
        if has:
            m, v = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
            dm, dv = torch.empty_like(dst), torch.empty_like(dst)
            new_states[name] = (dm, dv)
            keep_alive += [m, v]
        recs.append(_lib.DensifyAttr(src.data_ptr(), dst.data_ptr(), m.data_ptr() if has else None,
                                     dm.data_ptr() if has else None, v.data_ptr() if has else None,
                                     dv.data_ptr() if has else None, width))
        new_params[name] = dst
        keep_alive.append(src)
    arr = (_lib.DensifyAttr * len(recs))(*recs)
    _lib.check(lib.vidu4d_densify_apply(N, inc.data_ptr(), n_orig, n_clone, n_split, len(recs), arr, names.index("xyz"),
                                        names.index("scaling"), names.index("rotation"), draws.data_ptr(), scaled,
                                        src_row.data_ptr(), kind.data_ptr(), stream), "densify apply")
    # ---- re-key the optimizer exactly as _resize_groups does
    out = {}
    for name in names:
        group = groups[name]
        old = group["params"][0]
        st = gm.optimizer.state.pop(old, None)
        new = nn.Parameter(new_params[name].requires_grad_(True))
        if st is not None:
            if name in new_states:
                st["exp_avg"], st["exp_avg_sq"] = new_states[name]
            gm.optimizer.state[new] = st
        group["params"][0] = new
        out[name] = new
    gm._assign(out)
    gm.xyz_gradient_accum = torch.zeros(rows, 1, device=dev)
    gm.denom = torch.zeros(rows, 1, device=dev)
    gm.max_radii2D = torch.zeros(rows, device=dev)
