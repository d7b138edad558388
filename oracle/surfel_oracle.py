"""ctypes front-end of the CPU oracle (oracle/surfel_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (vidu4d_amd/) never imports this module.

The call sequence mirrors CudaRasterizer::Rasterizer::forward / ::backward
(/root/reference/gs/submodules/diff-surfel-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-342,
:346-448) stage by stage, and every intermediate array the reference keeps in its geometry /
binning / image state is returned so that tests can compare stage outputs, not only the image.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsurfel_oracle.so")
_lib = None

BLOCK = 16


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "surfel_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsurfel_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_inclusive_scan.restype = C.c_uint32
        _lib.oracle_higher_msb.restype = C.c_uint32
        _lib.oracle_set_threads.restype = C.c_int
    return _lib


def set_threads(n: int) -> int:
    return lib().oracle_set_threads(C.c_int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def higher_msb(n: int) -> int:
    return int(lib().oracle_higher_msb(C.c_uint32(n)))


def forward(means3D, opacities, scales, rotations, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy,
            sh_degree=0, shs=None, colors_precomp=None, stats=False):
    """Full reference forward on the CPU.  Returns a dict with the outputs
    (color (3,H,W), others (8,H,W), radii (P,)) and all intermediates."""
    L = lib()
    means3D = _f32(means3D, (-1, 3))
    P = means3D.shape[0]
    opacities = _f32(opacities, (P,))
    scales = _f32(scales, (P, 2))
    rotations = _f32(rotations, (P, 4))
    viewmatrix = _f32(viewmatrix, (16,))
    projmatrix = _f32(projmatrix, (16,))
    campos = _f32(campos, (3,))
    bg = _f32(bg, (3,))
    if (shs is None) == (colors_precomp is None):
        raise ValueError("exactly one of shs / colors_precomp")
    M = 0
    if shs is not None:
        shs = _f32(shs)
        M = shs.shape[1]
        shs = shs.reshape(P, M, 3)
    else:
        colors_precomp = _f32(colors_precomp, (P, 3))
    W, H = int(W), int(H)
    tanfovx, tanfovy = float(np.float32(tanfovx)), float(np.float32(tanfovy))

    st = dict(P=P, M=M, D=int(sh_degree), W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["transMat"] = np.zeros((P, 9), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["normal_opacity"] = np.zeros((P, 4), np.float32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    L.oracle_preprocess(
        C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M), _p(means3D), _p(scales), _p(rotations), _p(opacities),
        _p(shs), _p(colors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), C.c_int(W), C.c_int(H),
        C.c_float(tanfovx), C.c_float(tanfovy), _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]),
        _p(st["transMat"]), _p(st["rgb"]), _p(st["normal_opacity"]), _p(st["clamped"]), _p(st["tiles_touched"]))

    st["point_offsets"] = np.zeros(P, np.uint32)
    R = int(L.oracle_inclusive_scan(C.c_int(P), _p(st["tiles_touched"]), _p(st["point_offsets"]))) if P else 0
    st["num_rendered"] = R

    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    st["grid"] = (gx, gy)
    keys_u = np.zeros(R, np.uint64)
    vals_u = np.zeros(R, np.uint32)
    if P:
        L.oracle_emit_keys(C.c_int(P), _p(st["means2D"]), _p(st["depths"]), _p(st["point_offsets"]), _p(st["radii"]),
                           C.c_int(W), C.c_int(H), _p(keys_u), _p(vals_u))
    st["keys_unsorted"], st["values_unsorted"] = keys_u, vals_u
    bit = higher_msb(gx * gy)
    st["sort_bits"] = 32 + bit
    keys_s = np.zeros(R, np.uint64)
    vals_s = np.zeros(R, np.uint32)
    L.oracle_sort_pairs(C.c_uint32(R), _p(keys_u), _p(vals_u), _p(keys_s), _p(vals_s), C.c_int(32 + bit))
    st["point_list_keys"], st["point_list"] = keys_s, vals_s
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    L.oracle_tile_ranges(C.c_uint32(R), _p(keys_s), C.c_int(gx * gy), _p(st["ranges"]))

    st["final_T"] = np.zeros((3, H, W), np.float32)
    st["n_contrib"] = np.zeros((2, H, W), np.uint32)
    st["color"] = np.zeros((3, H, W), np.float32)
    st["others"] = np.zeros((8, H, W), np.float32)
    feats = colors_precomp if colors_precomp is not None else st["rgb"]
    pairs = np.zeros(2, np.uint64)
    L.oracle_render_forward(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(vals_s), _p(st["means2D"]), _p(feats),
                            _p(st["transMat"]), _p(st["normal_opacity"]), _p(bg), _p(st["final_T"]),
                            _p(st["n_contrib"]), _p(st["color"]), _p(st["others"]), _p(pairs) if stats else None)
    st["pairs_visited"], st["pairs_contributed"] = int(pairs[0]), int(pairs[1])
    st["_inputs"] = dict(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations,
                         viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos, bg=bg, shs=shs,
                         colors_precomp=colors_precomp)
    return st


def backward(st, dL_dcolor, dL_dothers):
    """Full reference backward on the CPU (Rasterizer::backward).  Returns the 8 tensors of
    RasterizeGaussiansBackwardCUDA (rasterize_points.cu:239) plus the internal dL_dnormal."""
    L = lib()
    P, M, D, W, H = st["P"], st["M"], st["D"], st["W"], st["H"]
    inp = st["_inputs"]
    dL_dcolor = _f32(dL_dcolor, (3, H, W))
    dL_dothers = _f32(dL_dothers, (8, H, W))
    feats = inp["colors_precomp"] if inp["colors_precomp"] is not None else st["rgb"]

    acc_T = np.zeros((P, 9), np.float64)
    acc_m2d = np.zeros((P, 3), np.float64)
    acc_n = np.zeros((P, 3), np.float64)
    acc_o = np.zeros((P,), np.float64)
    acc_c = np.zeros((P, 3), np.float64)
    L.oracle_render_backward(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(inp["bg"]),
                             _p(st["means2D"]), _p(st["normal_opacity"]), _p(st["transMat"]), _p(feats),
                             _p(st["final_T"]), _p(st["n_contrib"]), _p(dL_dcolor), _p(dL_dothers), _p(acc_T),
                             _p(acc_m2d), _p(acc_n), _p(acc_o), _p(acc_c))
    g = dict(
        dL_dtransMat=acc_T.astype(np.float32), dL_dmeans2D=acc_m2d.astype(np.float32),
        dL_dnormal=acc_n.astype(np.float32), dL_dopacity=acc_o.astype(np.float32).reshape(P, 1),
        dL_dcolors=acc_c.astype(np.float32))
    g["dL_dmeans2D_filter"] = g["dL_dmeans2D"].copy()  # before the densification "hack" overwrites it
    g["dL_dtransMat_render"] = g["dL_dtransMat"].copy()
    g.update(backward_chain(st, g["dL_dtransMat_render"], g["dL_dmeans2D_filter"], g["dL_dnormal"], g["dL_dcolors"]))
    return g


def backward_chain(st, dL_dtransMat_render, dL_dmeans2D_filter, dL_dnormal, dL_dcolors, after_aabb=False):
    """The per-surfel chain rule behind the blend (BACKWARD::preprocess, backward.cu:601-668: the AABB's share of
    dL_dtransMat / dL_dmeans2D, then transMat -> mean / scale / rotation and colour -> SH) from the blend's accumulators.
    Split out of backward() so a test can push ANOTHER implementation's blend sums through it and compare that
    implementation's chain with this one on equal inputs (tools/fuzz_footprint_gpu.py: the chain amplifies rounding
    differences of the sums by two orders of magnitude).  after_aabb: dL_dtransMat already holds the AABB's share (the
    tensor RasterizeGaussiansBackwardCUDA returns); dL_dmeans2D_filter may then be None."""
    L = lib()
    P, M, D, W, H = st["P"], st["M"], st["D"], st["W"], st["H"]
    inp = st["_inputs"]
    g = dict(dL_dtransMat=_f32(dL_dtransMat_render, (P, 9)).copy())
    if not after_aabb:
        g["dL_dmeans2D"] = _f32(dL_dmeans2D_filter, (P, 3)).copy()
    dL_dnormal = _f32(dL_dnormal, (P, 3))
    dL_dcolors = _f32(dL_dcolors, (P, 3))
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(st["tanfovy"]))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(st["tanfovx"]))
    Wh = np.float32(focal_x * np.float32(st["tanfovx"]))
    Hh = np.float32(focal_y * np.float32(st["tanfovy"]))
    if not after_aabb:
        L.oracle_aabb_backward(C.c_int(P), _p(st["radii"]), C.c_float(float(Wh)), C.c_float(float(Hh)),
                               _p(st["transMat"]), _p(g["dL_dmeans2D"]), _p(g["dL_dtransMat"]))

    g["dL_dsh"] = np.zeros((P, M, 3), np.float32)
    g["dL_dmeans3D"] = np.zeros((P, 3), np.float32)
    g["dL_dscales"] = np.zeros((P, 2), np.float32)
    g["dL_drotations"] = np.zeros((P, 4), np.float32)
    L.oracle_preprocess_backward(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(inp["means3D"]), _p(st["radii"]), _p(inp["shs"]), _p(st["clamped"]),
        _p(inp["scales"]), _p(inp["rotations"]), _p(inp["viewmatrix"]), C.c_float(float(focal_x)),
        C.c_float(float(focal_y)), C.c_float(st["tanfovx"]), C.c_float(st["tanfovy"]), _p(inp["campos"]),
        _p(g["dL_dtransMat"]), _p(dL_dnormal), _p(dL_dcolors), _p(g["dL_dsh"]), _p(g["dL_dmeans3D"]),
        _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix):
    means3D = _f32(means3D, (-1, 3))
    present = np.zeros(means3D.shape[0], np.uint8)
    lib().oracle_mark_visible(C.c_int(means3D.shape[0]), _p(means3D), _p(_f32(viewmatrix, (16,))), _p(present))
    return present.astype(bool)


def distortion_f64(st):
    """Plane 6 in fp64 from the oracle's own fp32 per-pair alpha / depth and walk (oracle_distortion_f64): the yardstick for
    "which fp32 formulation of the distortion is closer to what it means" (tests/test_gpu_reference.py)."""
    out = np.zeros((st["H"], st["W"]), np.float64)
    lib().oracle_distortion_f64(C.c_int(st["W"]), C.c_int(st["H"]), _p(st["ranges"]), _p(st["point_list"]), _p(st["means2D"]),
                                _p(st["transMat"]), _p(st["normal_opacity"]), _p(out))
    return out
