"""ctypes front-end of oracle/_ref/libref_surfel.so: the reference's own rasterizer sources compiled
for gfx950 (oracle/ref_build/build_ref.py).  TEST INFRASTRUCTURE; needs a GPU; never imported by the
product.  Inputs/outputs are torch tensors on the GPU."""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBS = {"default": os.path.join(os.path.dirname(_HERE), "_ref", "libref_surfel.so"),
        # fp contraction off, rsqrtf = 1/sqrtf: the reference's arithmetic in source order (build_ref.py)
        "strict": os.path.join(os.path.dirname(_HERE), "_ref", "libref_surfel_strict.so")}
LIB = LIBS["default"]
_libs = {}
_variant = "default"


def available(variant: str = "default") -> bool:
    return os.path.exists(LIBS[variant])


def use(variant: str):
    """Selects the build the following forward / backward / state calls run ("default" or "strict")."""
    global _variant
    assert variant in LIBS, variant
    _variant = variant


def lib():
    if _variant not in _libs:
        L = C.CDLL(LIBS[_variant])
        L.ref_forward.restype = C.c_int
        L.ref_state.restype = C.c_long
        _libs[_variant] = L
    return _libs[_variant]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


def forward(sc, colors_precomp=None):
    """sc: SurfelScene on the GPU.  Returns dict(color, others, radii, num_rendered) (GPU tensors)."""
    L = lib()
    dev = sc.means3D.device
    P, W, H = sc.num_surfels, sc.width, sc.height
    color = torch.zeros(3, H, W, device=dev)
    others = torch.zeros(8, H, W, device=dev)
    radii = torch.zeros(P, dtype=torch.int32, device=dev)
    shs = None if colors_precomp is not None else sc.shs.contiguous()
    M = 0 if shs is None else shs.shape[1]
    torch.cuda.synchronize()
    R = L.ref_forward(C.c_int(P), C.c_int(sc.sh_degree), C.c_int(M), _p(sc.bg), C.c_int(W), C.c_int(H),
                      _p(sc.means3D), _p(shs), _p(colors_precomp), _p(sc.opacities), _p(sc.scales), _p(sc.rotations),
                      _p(sc.viewmatrix), _p(sc.projmatrix), _p(sc.campos), C.c_float(sc.tanfovx),
                      C.c_float(sc.tanfovy), _p(color), _p(others), _p(radii))
    return dict(color=color, others=others, radii=radii, num_rendered=R, shs=shs, colors_precomp=colors_precomp)


def backward(sc, fwd, dL_dcolor, dL_dothers):
    L = lib()
    dev = sc.means3D.device
    P, W, H = sc.num_surfels, sc.width, sc.height
    M = 0 if fwd["shs"] is None else fwd["shs"].shape[1]
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731  (the reference accumulates into zero-filled outputs)
    g = dict(dL_dmeans2D=z(P, 3), dL_dnormal=z(P, 3), dL_dopacity=z(P, 1), dL_dcolors=z(P, 3), dL_dmeans3D=z(P, 3),
             dL_dtransMat=z(P, 9), dL_dsh=z(P, M, 3), dL_dscales=z(P, 2), dL_drotations=z(P, 4))
    torch.cuda.synchronize()
    L.ref_backward(C.c_int(P), C.c_int(sc.sh_degree), C.c_int(M), _p(sc.bg), C.c_int(W), C.c_int(H), _p(sc.means3D),
                   _p(fwd["shs"]), _p(fwd["colors_precomp"]), _p(sc.scales), _p(sc.rotations), _p(sc.viewmatrix),
                   _p(sc.projmatrix), _p(sc.campos), C.c_float(sc.tanfovx), C.c_float(sc.tanfovy), _p(fwd["radii"]),
                   _p(dL_dcolor.contiguous()), _p(dL_dothers.contiguous()), _p(g["dL_dmeans2D"]), _p(g["dL_dnormal"]),
                   _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dtransMat"]),
                   _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


_STATE = dict(point_list=(0, np.uint32), sorted_keys=(1, np.uint64), ranges=(2, np.uint32), n_contrib=(3, np.uint32),
              final_T=(4, np.float32), transMat=(5, np.float32), means2D=(6, np.float32), depths=(7, np.float32),
              rgb=(8, np.float32), normal_opacity=(9, np.float32), tiles_touched=(10, np.uint32))


def state(name, max_elems):
    what, dt = _STATE[name]
    buf = np.zeros(max(max_elems, 1), dt)
    n = lib().ref_state(C.c_int(what), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
    assert n >= 0, (name, n)
    return buf[: n // buf.itemsize]
