"""A time-conditioned network's dense layers on a handful of rows as ONE launch per direction (csrc/dense_stack.hip).

    dense_stack(x (R, 256), trunk, head_a, head_b) -> (out_a (R, Oa), out_b (R, Ob))

`trunk`, `head_a`, `head_b` are lists of (nn.Linear, relu: bool, scale: float) -- the layers of a nets.TimeMLP (reference
lab4d/nnutils/time.py:11-133: linear_1..linear_D + linear_final, each followed by its ReLU) and of the two heads on its
features (pose.py:29-150 CameraMLP.trans / .quat, :153-323 ArticulationFlatMLP.trans (ScaleLayer 0.1) / .so3).  Values as
F.linear / F.relu to float rounding; differentiable once w.r.t. x and every weight and bias."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .nets import ScaleLayer


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def sequential_layers(seq) -> list | None:
    """A Sequential of Linear / ReLU / ScaleLayer (or a bare Linear) as [(linear, relu, scale)], None if it holds anything else."""
    mods = [seq] if isinstance(seq, nn.Linear) else list(seq)
    out = []
    for m in mods:
        if isinstance(m, nn.Linear):
            out.append([m, False, 1.0])
        elif isinstance(m, nn.ReLU) and out and not out[-1][1] and out[-1][2] == 1.0:
            out[-1][1] = True
        elif isinstance(m, ScaleLayer) and out:
            sc = float(m.scale_value) if hasattr(m, "scale_value") else float(m.scale.item())
            # (the kernels' backward reads the ReLU's mask off the stored, SCALED activation, h_out > 0: right only for a
            # positive scale behind a ReLU -- ADVICE r5; anything else goes to the library layers)
            if out[-1][1] and not sc > 0.0:
                return None
            out[-1][2] *= sc
        else:
            return None
    return [tuple(x) for x in out]


def _describe(rows, groups, tensors, grads=None):
    """groups: three lists of (relu, scale); tensors: flat [W0, b0, W1, b1, ...] in the same order."""
    d = _lib.DenseStack()
    d.rows, d.n_trunk, d.n_head_a, d.n_head_b = rows, len(groups[0]), len(groups[1]), len(groups[2])
    l = 0
    for g in groups:
        for relu, scale in g:
            W, b = tensors[2 * l], tensors[2 * l + 1]
            d.in_[l], d.out[l], d.relu[l], d.scale[l] = W.shape[1], W.shape[0], int(relu), float(scale)
            d.W[l], d.b[l] = W.data_ptr(), (b.data_ptr() if b is not None else None)
            if grads is not None:
                d.gW[l], d.gb[l] = grads[2 * l].data_ptr(), (grads[2 * l + 1].data_ptr() if b is not None else None)
            l += 1
    return d


class _DenseStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, groups, *tensors):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("dense_stack: float32 HIP tensors required")
        x = x.detach().contiguous()
        ts = [None if t is None else t.detach().contiguous() for t in tensors]
        R = x.shape[0]
        desc = _describe(R, groups, ts)
        lib = _lib.load()
        n = lib.vidu4d_dense_stack_acts_floats(C.byref(desc))
        if n < 0:
            raise RuntimeError("dense_stack: unsupported stack (rows <= 16, widths <= 256, consistent layer shapes)")
        acts = torch.empty(n, device=x.device)
        _lib.check(lib.vidu4d_dense_stack_forward(C.byref(desc), x.data_ptr(), acts.data_ptr(), _stream(x)), "dense_stack forward")
        ctx.groups = groups
        ctx.save_for_backward(x, acts, *[t for t in ts if t is not None])
        ctx.has = [t is not None for t in ts]
        n_layers = len(ts) // 2
        outs_l = [len(groups[0]) + len(groups[1]) - 1, n_layers - 1] if groups[1] and groups[2] else [n_layers - 1]
        offs, off = [], 0
        for l in range(n_layers):
            offs.append(off)
            off += R * ts[2 * l].shape[0]
        res = tuple(acts[offs[l]:offs[l] + R * ts[2 * l].shape[0]].view(R, ts[2 * l].shape[0]) for l in outs_l)
        return res if len(res) > 1 else res[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        x, acts, *rest = ctx.saved_tensors
        it = iter(rest)
        ts = [next(it) if h else None for h in ctx.has]
        # every weight / bias gradient in ONE allocation (the kernel writes all of it)
        sizes = [0 if t is None else t.numel() for t in ts]
        flat = torch.empty(sum(sizes), device=x.device)
        grads, off = [], 0
        for t, sz in zip(ts, sizes):
            grads.append(None if t is None else flat[off:off + sz].view_as(t))
            off += sz
        desc = _describe(x.shape[0], ctx.groups, ts, grads)
        g = [None if t is None else t.detach().float().contiguous() for t in gouts]
        ga = g[0]
        gb = g[1] if len(g) > 1 else None
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = torch.empty_like(acts)   # (every layer's g_pre, between the backward's two launches)
        lib = _lib.load()
        _lib.check(lib.vidu4d_dense_stack_backward(C.byref(desc), x.data_ptr(), acts.data_ptr(),
                                                   None if ga is None else ga.data_ptr(), None if gb is None else gb.data_ptr(),
                                                   ws.data_ptr(), None if gx is None else gx.data_ptr(),
                                                   _stream(x)), "dense_stack backward")
        return (gx, None) + tuple(grads)


def dense_stack(x, trunk, head_a=(), head_b=()):
    """trunk / head_a / head_b: lists of (nn.Linear, relu: bool, scale: float).  Returns the heads' outputs (or the trunk's
    when there are no heads)."""
    groups = tuple(tuple((bool(r), float(s)) for _, r, s in g) for g in (trunk, head_a, head_b))
    tensors = []
    for g in (trunk, head_a, head_b):
        for lin, _, _ in g:
            tensors += [lin.weight, lin.bias]
    return _DenseStack.apply(x, groups, *tensors)


def supported(x, *groups) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[0] <= _lib.DenseStack.MAX_ROWS):
        return False
    n = sum(len(g) for g in groups)
    if n > _lib.DenseStack.MAX_LAYERS or not groups[0]:
        return False
    return all(lin.weight.shape[0] <= 256 and lin.weight.shape[1] <= 256 for g in groups for lin, _, _ in g)
