"""ctypes front-end of tests/host_emul (TEST INFRASTRUCTURE: the product's arithmetic header run on the host)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhost_emul.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "host_emul.cpp")
        hdr = os.path.join(_HERE, "..", "..", "vidu4d_amd", "csrc", "surfel_math.h")
        if (not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-mfma",
                                   "-Wno-unknown-pragmas", "-o", _LIB, src])
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run(st, dL_dcolor, dL_dothers, cull=True, lite=False):
    """Runs the emulated pipeline on the oracle's inputs / binning (st = oracle forward state)."""
    L = lib()
    inp = st["_inputs"]
    P, D, M, W, H = st["P"], st["D"], st["M"], st["W"], st["H"]
    out = {}
    out["radii"] = np.zeros(P, np.int32)
    out["tiles"] = np.zeros(P, np.uint32)
    out["rec"] = np.zeros((P, 32), np.float32)
    L.emul_preprocess(C.c_int(P), C.c_int(D), C.c_int(M), _p(inp["means3D"]), _p(inp["scales"]), _p(inp["rotations"]),
                      _p(inp["opacities"]), _p(inp["shs"]), _p(inp["colors_precomp"]), _p(inp["viewmatrix"]),
                      _p(inp["campos"]), C.c_int(W), C.c_int(H), C.c_float(st["tanfovx"]), C.c_float(st["tanfovy"]),
                      _p(out["radii"]), _p(out["tiles"]), _p(out["rec"]))
    out["final_T"] = np.zeros((3, H, W), np.float32)
    out["n_contrib"] = np.zeros((2, H, W), np.uint32)
    out["color"] = np.zeros((3, H, W), np.float32)
    out["others"] = np.zeros((8, H, W), np.float32)
    L.emul_render_fwd(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(out["rec"]), _p(inp["bg"]),
                      _p(out["final_T"]), _p(out["n_contrib"]), _p(out["color"]), _p(out["others"]), C.c_int(int(cull)),
                      C.c_int(int(lite)))
    acc = np.zeros((P, 20), np.float64)
    dc = np.ascontiguousarray(dL_dcolor, np.float32)
    do = np.ascontiguousarray(dL_dothers, np.float32)
    L.emul_render_bwd(C.c_int(W), C.c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(out["rec"]), _p(inp["bg"]),
                      _p(out["final_T"]), _p(out["n_contrib"]), _p(dc), _p(do), _p(acc), C.c_int(int(cull)), C.c_int(int(lite)))
    accf = acc.astype(np.float32)
    out["acc"] = accf
    g = dict(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32),
             dL_dcolors=np.zeros((P, 3), np.float32), dL_dopacity=np.zeros((P, 1), np.float32),
             dL_dtransMat=np.zeros((P, 9), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
             dL_dscales=np.zeros((P, 2), np.float32), dL_drotations=np.zeros((P, 4), np.float32))
    L.emul_preprocess_bwd(C.c_int(P), C.c_int(D), C.c_int(M), _p(inp["means3D"]), _p(inp["scales"]),
                          _p(inp["rotations"]), _p(inp["shs"]), _p(inp["viewmatrix"]), _p(inp["campos"]), C.c_int(W),
                          C.c_int(H), C.c_float(st["tanfovx"]), C.c_float(st["tanfovy"]), _p(out["radii"]),
                          _p(out["rec"]), _p(accf), _p(g["dL_dmeans3D"]), _p(g["dL_dmeans2D"]), _p(g["dL_dcolors"]),
                          _p(g["dL_dopacity"]), _p(g["dL_dtransMat"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]),
                          _p(g["dL_drotations"]))
    out["grads"] = g
    return out


def footprint_scan(st):
    """Per (visible surfel, 8x8 quadrant): the rectangle form of the footprint test the blend kernels run against the exact
    pair test and the per-pixel form, over the whole image -> counts (see host_emul.cpp: emul_footprint_scan)."""
    L = lib()
    inp = st["_inputs"]
    P, D, M, W, H = st["P"], st["D"], st["M"], st["W"], st["H"]
    radii, tiles, rec = np.zeros(P, np.int32), np.zeros(P, np.uint32), np.zeros((P, 32), np.float32)
    L.emul_preprocess(C.c_int(P), C.c_int(D), C.c_int(M), _p(inp["means3D"]), _p(inp["scales"]), _p(inp["rotations"]),
                      _p(inp["opacities"]), _p(inp["shs"]), _p(inp["colors_precomp"]), _p(inp["viewmatrix"]),
                      _p(inp["campos"]), C.c_int(W), C.c_int(H), C.c_float(st["tanfovx"]), C.c_float(st["tanfovy"]),
                      _p(radii), _p(tiles), _p(rec))
    counts = np.zeros(5, np.int64)
    L.emul_footprint_scan(C.c_int(P), C.c_int(W), C.c_int(H), _p(radii), _p(rec), _p(counts))
    return dict(kept=int(counts[0]), contributing=int(counts[1]), dropped_contributing=int(counts[2]),
                kept_between_pixel_centres=int(counts[3]), box_would_keep=int(counts[4]))
