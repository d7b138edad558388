"""Feature-major (W @ X, outputs (out, N)) against row-major (X @ W^T, outputs (N, out)) for the delta-skin MLP's shapes at
N = 200 000 (75 -> 64 -> 64 -> 25), forward + backward through autograd, library GEMMs.  GPU box."""
import time, torch
dev = torch.device("cuda:0")
N = 200000
torch.manual_seed(0)
W1, W2, W3 = (torch.randn(64, 75, device=dev, requires_grad=True), torch.randn(64, 64, device=dev, requires_grad=True), torch.randn(25, 64, device=dev, requires_grad=True))
b1, b2, b3 = (torch.randn(64, device=dev, requires_grad=True), torch.randn(64, device=dev, requires_grad=True), torch.randn(25, device=dev, requires_grad=True))
A, c0 = torch.randn(75, 3, device=dev, requires_grad=True), torch.randn(75, device=dev, requires_grad=True)
xyz = torch.randn(N, 3, device=dev, requires_grad=True)
g_x, g_r = torch.randn(75, N, device=dev), torch.randn(25, N, device=dev)

def feature_major():
    xbT = torch.addmm(c0[:, None], A, xyz.t())
    h = torch.relu(torch.addmm(b1[:, None], W1, xbT))
    h = torch.relu(torch.addmm(b2[:, None], W2, h))
    rawT = torch.addmm(b3[:, None], W3, h)
    return xbT, rawT

def row_major():
    xb = torch.addmm(c0, xyz, A.t())
    h = torch.relu(torch.nn.functional.linear(xb, W1, b1))
    h = torch.relu(torch.nn.functional.linear(h, W2, b2))
    raw = torch.nn.functional.linear(h, W3, b3)
    return xb.t().contiguous(), raw.t().contiguous()

for name, fn in (("feature-major", feature_major), ("row-major + transposes", row_major)):
    for _ in range(3):
        xbT, rawT = fn(); torch.autograd.backward([xbT, rawT], [g_x, g_r])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        xbT, rawT = fn(); torch.autograd.backward([xbT, rawT], [g_x, g_r])
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms fwd+bwd")
a = feature_major(); b = row_major()
print("max |diff| xbT", float((a[0] - b[0]).abs().max()), "rawT", float((a[1] - b[1]).abs().max()), "scale", float(a[1].abs().max()))
