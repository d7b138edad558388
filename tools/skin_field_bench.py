"""Times csrc/skin_field.hip alone (GPU box): forward and backward of the bob delta-skin field at the fit step's shape
(200k surfels, 25 bones, D = 2 hidden layers of width 64), HIP events around 20 launches each.
Usage: python tools/skin_field_bench.py [N] [B] [D]"""
import os, sys
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vidu4d_amd import _lib
from vidu4d_amd.lab4d.lbs_fused import _skin_field_args, pack_skin_field

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
lim = _lib.SKIN_FIELD
W, IN, OUT = lim["width"], lim["in_max"], lim["out_max"]
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
tab = {"B": B, "D": D, "bone_A": r(3 * B, 3), "bone_c": r(3 * B), "w_in": torch.zeros(W, IN, device=dev), "w_out": torch.zeros(OUT, W, device=dev),
       "b_out": torch.zeros(OUT, device=dev)}
tab["w_in"][:, :3 * B] = 0.2 * r(W, 3 * B)
tab["w_out"][:B] = 0.2 * r(B, W)
tab["b_out"][:B] = r(B)
if D > 1:
    tab["w_hid"], tab["b_hid"] = 0.2 * r(D - 1, W, W), r(D - 1, W)
xyz, b_in = 0.3 * r(N, 3), r(W)
rawT = torch.empty(B, N, device=dev)
xbT = torch.empty(3 * B, N, device=dev)
masks = torch.empty(D * 64 * ((N + 31) // 32), dtype=torch.int32, device=dev)
g_rawT, g_xbT, g_xyz = r(B, N), r(3 * B, N), torch.empty(N, 3, device=dev)
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
if os.environ.get("SKIN_PACK", "1") != "0":
    pack_skin_field(tab)
print("weights:", "packed image" if "packed_fwd" in tab else "gathered per launch")
cases = {
    "forward (rawT only, as the fit step)": (lib.vidu4d_skin_field_forward, _skin_field_args(tab, N, xyz, b_in, rawT=rawT, relu_masks=masks)),
    "forward (+ xbT)": (lib.vidu4d_skin_field_forward, _skin_field_args(tab, N, xyz, b_in, xbT=xbT, rawT=rawT, relu_masks=masks)),
    "backward (g_rawT + g_xbT, masks from the forward)": (lib.vidu4d_skin_field_backward,
                                                           _skin_field_args(tab, N, xyz, b_in, g_xbT=g_xbT, g_rawT=g_rawT, g_xyz=g_xyz, relu_masks=masks)),
    "backward (recomputing the forward)": (lib.vidu4d_skin_field_backward,
                                           _skin_field_args(tab, N, xyz, b_in, g_xbT=g_xbT, g_rawT=g_rawT, g_xyz=g_xyz)),
}
tiles = (N + 31) // 32
T1 = (3 * B + 1) // 2
mf_fwd = tiles * (2 * T1 + 64 * (D - 1) + 32)
mf_bwd = tiles * (2 * ((B + 1) // 2) + 64 * (D - 1) + 96)   # output gradient, hidden layers, first layer's three row blocks
for name, (fn, a) in cases.items():
    for _ in range(3):
        _lib.check(fn(a, st), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(a, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mf = mf_fwd if name.startswith("forward") else (mf_bwd + (mf_fwd - tiles * 32 if "recomputing" in name else 0))  # MFMAs of the launch
    # v_mfma_f32_32x32x2_f32: 64 cycles of a SIMD's matrix pipe each; 1024 SIMDs at 2.4 GHz
    print(f"{name:52s} {us:7.1f} us   MFMA pipe time {mf * 64 / 1024 / 2.4e3:6.1f} us ({100 * mf * 64 / 1024 / 2.4e3 / us:4.1f} % busy)")
