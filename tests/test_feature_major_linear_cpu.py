"""bob_warp.feature_major_linear: W @ X + b over N surfels whose weight gradient is cut along the surfels into a batched
GEMM (the library's kernel for a 64 x 75 x 200 000 contraction occupies a handful of workgroups).  Same sums, another order."""
import pytest
import torch

from vidu4d_amd.lab4d.bob_warp import SPLIT_K_CHUNK, contract_over_columns, feature_major_linear


@pytest.mark.parametrize("O,I,N,chunk", [(64, 75, 4 * SPLIT_K_CHUNK + 5, SPLIT_K_CHUNK), (25, 64, 9000, 1024), (64, 64, 8192, 2048),
                                         (7, 3, 100, 2048), (64, 75, 20001, 512)])
def test_split_contraction_equals_the_plain_one(O, I, N, chunk):
    g = torch.Generator().manual_seed(N)
    G, X = torch.randn(O, N, generator=g), torch.randn(I, N, generator=g)
    want = (G.double() @ X.double().t())
    got = contract_over_columns(G, X, chunk)
    assert got.shape == (O, I)
    Xt = X.t().contiguous().t()        # (I, N) as the transpose of row-major points: the bone map's input
    assert torch.equal(contract_over_columns(G, Xt, chunk), got) or float((contract_over_columns(G, Xt, chunk) - got).abs().max()) <= 1e-4
    assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max()) * 8


def test_linear_gradients_equal_addmm(monkeypatch):
    g = torch.Generator().manual_seed(0)
    N = 4 * SPLIT_K_CHUNK + 77
    X0 = torch.randn(75, N, generator=g)
    res = {}
    for split in (True, False):
        W = torch.randn(64, 75, generator=torch.Generator().manual_seed(1)).requires_grad_()
        b = torch.randn(64, generator=torch.Generator().manual_seed(2)).requires_grad_()
        X = X0.clone().requires_grad_()
        Y = feature_major_linear(b, W, X, SPLIT_K_CHUNK) if split else torch.addmm(b[:, None], W, X)
        Y.backward(torch.randn(Y.shape, generator=torch.Generator().manual_seed(3)))
        res[split] = (Y.detach(), W.grad, b.grad, X.grad)
    for a, c in zip(res[True], res[False]):
        assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())
    # frozen weights: the plain library call, no autograd node of ours
    W = torch.randn(64, 75)
    assert feature_major_linear(torch.zeros(64), W, X0).grad_fn is None
