"""The networks' forward and backward for a step's frames as two captured hipGraphs whose parameter gradients LAND IN
`.grad` WITHOUT A COPY.

`torch.cuda.make_graphed_callables` (round 5's first form) returns the captured backward's gradients to autograd: one
AccumulateGrad engine node per parameter tensor and step (66 tensors), a copy wherever a static buffer cannot be adopted,
and the trainer's round accumulation copied each again -- together + 20 % of the step in the same-process A/B
(`graphed_warp_networks: "torch"`, `profiles/r05_fit_optim_warp_ab.txt`).  Here the
backward replays the graph and hands every parameter its static gradient buffer directly -- `p.grad = buffer` when the
parameter has none (the usual case: `zero_grad(set_to_none=True)`), `p.grad += buffer` otherwise; a `.grad` that still IS
the buffer from an earlier backward (gradient accumulation over several backward calls) is detached from it first.
Consequences, stated: gradients of these parameters arrive through `.backward()` only (`torch.autograd.grad` w.r.t. them
returns None), tensor hooks on them do not fire, and a gradient stays valid until the next backward through the same graphs.
The module is called with integer frame ids only; its outputs are float tensors of fixed shapes.

ONE forward per backward: the outputs are the graphs' static buffers, which whoever consumes them saves for its own
backward (lbs_fused.lbs_skin_apply); a second replay -- grad-enabled or not -- before the first one's backward overwrites them
through raw pointers that autograd's version counters never see.  Every replay therefore advances `generation`, a grad-enabled
forward remembers its own, and its backward RAISES when another replay came in between instead of returning gradients of the
wrong frames (Stage3Trainer makes exactly one forward per backward; a caller that needs more sets
`graphed_warp_networks: False`)."""
from __future__ import annotations

import torch


class GraphedNetworks:
    def __init__(self, module: torch.nn.Module, frame_id: torch.Tensor, params, warmup: int = 3):
        self.params = [p for p in params if p.requires_grad]
        dev = frame_id.device
        self.static_in = frame_id.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # lazily built tables, library handles and workspaces: before any capture
            for _ in range(warmup):
                outs = module(self.static_in)
                torch.autograd.grad(outs, self.params, [torch.ones_like(o) for o in outs], allow_unused=True)
            del outs
        torch.cuda.current_stream(dev).wait_stream(side)
        pool = torch.cuda.graph_pool_handle()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, pool=pool):
            self.static_outs = tuple(module(self.static_in))
        self.static_gouts = tuple(torch.zeros_like(o) for o in self.static_outs)
        with torch.cuda.graph(self.bwd_graph, pool=pool):
            self.static_grads = torch.autograd.grad(self.static_outs, self.params, self.static_gouts, allow_unused=True)
        self.live = [(p, g) for p, g in zip(self.params, self.static_grads) if g is not None]
        # (the autograd graph of the captured forward is not needed again: keeping it would keep the parameters' AccumulateGrad
        # nodes of the CAPTURE stream alive, which later backwards through the same parameters outside the graphs would find)
        self.static_outs = tuple(o.detach() for o in self.static_outs)
        self.generation = 0   # replays so far (see the module docstring)
        owner = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, frame_id, anchor):
                owner.static_in.copy_(frame_id)
                owner.fwd_graph.replay()
                owner.generation += 1
                ctx.generation = owner.generation
                return tuple(o.detach() for o in owner.static_outs)

            @staticmethod
            @torch.autograd.function.once_differentiable
            def backward(ctx, *gouts):
                if ctx.generation != owner.generation:
                    raise RuntimeError("GraphedNetworks: the networks were evaluated again (%d replay(s)) between this forward "
                                       "and its backward; their outputs are the captured graphs' static buffers, so the tensors "
                                       "saved for this backward have been overwritten.  One forward per backward, or set "
                                       "graphed_warp_networks=False." % (owner.generation - ctx.generation))
                for s, g in zip(owner.static_gouts, gouts):
                    if g is None:
                        s.zero_()
                    else:
                        s.copy_(g)
                for p, g in owner.live:   # (an accumulated gradient that aliases the buffer the replay overwrites)
                    if p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                        p.grad = p.grad.clone()
                owner.bwd_graph.replay()
                for p, g in owner.live:
                    if p.grad is None:
                        p.grad = g
                    else:
                        p.grad.add_(g)
                return None, None

        self._fn = _Replay
        # (autograd only runs a Function's backward if an input requires grad: a scalar that does, and gets None back)
        self._anchor = torch.zeros((), device=dev, requires_grad=True)

    def __call__(self, frame_id: torch.Tensor):
        if torch.is_grad_enabled() and self.live:
            return self._fn.apply(frame_id, self._anchor)
        self.static_in.copy_(frame_id)
        self.fwd_graph.replay()
        self.generation += 1   # (a pending grad-enabled forward's saved outputs are gone: its backward will say so)
        return tuple(o.detach() for o in self.static_outs)
