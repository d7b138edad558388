"""csrc/dense_stack.hip: a TimeMLP's layers and its two heads on a handful of rows in one launch per direction, against
the same layers as torch library calls (values and every gradient)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _grads(mods, x):
    return [x.grad.clone()] + [None if p.grad is None else p.grad.clone() for m in mods for p in m.parameters()]


@pytest.mark.parametrize("rows", [1, 3, 4, 9, 16])
@pytest.mark.parametrize("net", ["articulation", "camera"])
def test_time_mlp_heads_match_the_library_layers(gpu_device, rows, net):
    from vidu4d_amd.lab4d.nets import ArticulationFlatMLP, CameraMLP, make_frame_info
    dev = gpu_device
    torch.manual_seed(rows)
    info = make_frame_info([0, 12])
    if net == "articulation":
        mlp = ArticulationFlatMLP(info, num_se3=25).to(dev)
        heads = (mlp.so3, mlp.trans)
    else:
        mlp = CameraMLP(torch.eye(4).repeat(12, 1, 1), frame_info=info).to(dev)
        heads = (mlp.quat, mlp.trans)
    with torch.no_grad():   # (weights of a size that lets every layer's gradient matter)
        for p in mlp.parameters():
            p.add_(0.03 * torch.randn_like(p))
    x0 = torch.randn(rows, 256, device=dev)
    gouts = None
    res = {}
    for fused in (True, False):
        mlp.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        if fused:
            out = mlp.fused_heads(x, *heads)
            assert out is not None, "the stack was not taken by the kernel"
        else:
            feat = mlp.features(x)
            out = heads[0](feat), heads[1](feat)
        if gouts is None:
            gouts = [torch.randn_like(o) for o in out]
        torch.autograd.backward(out, gouts)
        res[fused] = ([o.detach() for o in out], _grads([mlp], x))
    for a, b in zip(res[True][0], res[False][0]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-3, float(b.abs().max()))
    names = ["x"] + [k for k, _ in mlp.named_parameters()]
    used = 0
    for k, a, b in zip(names, res[True][1], res[False][1]):
        assert (a is None) == (b is None), k
        if b is None:   # (the time embedding's own layers: not part of the stack)
            continue
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-9, k
        used += float(b.abs().max()) > 0
    assert used >= 20


def test_plain_stack_with_odd_widths(gpu_device):
    """No heads, widths that do not divide the block, a layer without bias, a scaled last layer: the kernels' general paths."""
    from vidu4d_amd.lab4d.dense_stack import dense_stack
    dev = gpu_device
    torch.manual_seed(0)
    lins = [nn.Linear(100, 37).to(dev), nn.Linear(37, 256, bias=False).to(dev), nn.Linear(256, 5).to(dev)]
    spec = [(lins[0], True, 1.0), (lins[1], True, 1.0), (lins[2], False, 0.25)]
    x0 = torch.randn(7, 100, device=dev)
    g = torch.randn(7, 5, device=dev)
    res = {}
    for fused in (True, False):
        for l in lins:
            l.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        y = dense_stack(x, spec) if fused else 0.25 * lins[2](torch.relu(lins[1](torch.relu(lins[0](x)))))
        y.backward(g)
        res[fused] = (y.detach(), _grads(lins, x))
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5 * float(res[False][0].abs().max())
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
