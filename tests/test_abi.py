"""CPU tests of the C-ABI boundary: the shared object builds for gfx950, loads without a GPU,
exports every symbol include/vidu4d_surfel.h declares, and validates arguments before touching
the device."""
import ctypes as C
import os
import re

import pytest

from vidu4d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "vidu4d_surfel.h")).read()
    diag = open(os.path.join(ROOT, "include", "vidu4d_surfel_diag.h")).read()  # (diagnostics: not the drop-in boundary)
    assert not re.findall(r"\b(vidu4d_surfel_profile_[a-z_]+)\s*\(", hdr), "diagnostics belong in vidu4d_surfel_diag.h"
    declared = set(re.findall(r"\b(vidu4d_[a-z0-9_]+)\s*\(", hdr + diag))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.vidu4d_surfel_abi_version() == _lib.ABI_VERSION


def test_struct_layout_matches_header():
    """ctypes mirrors must have the field order of the C structs."""
    hdr = open(os.path.join(ROOT, "include", "vidu4d_surfel.h")).read()
    for cname, struct in (("Vidu4dSurfelForwardArgs", _lib.ForwardArgs), ("Vidu4dSurfelBackwardArgs", _lib.BackwardArgs),
                          ("Vidu4dAdamTensor", _lib.AdamTensor), ("Vidu4dDensifyAttr", _lib.DensifyAttr),
                          ("Vidu4dSkinFieldArgs", _lib.SkinFieldArgs), ("Vidu4dStage3LossArgs", _lib.Stage3LossArgs),
                          ("Vidu4dStage3LossGrads", _lib.Stage3LossGrads)):
        body = hdr[hdr.index("typedef struct " + cname):hdr.index("} " + cname + ";")]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split("{", 1)[1].split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", re.sub(r"\[.*?\]", "", part).strip())[0])
        assert names == [f[0] for f in struct._fields_], cname


def test_sizes_monotonic():
    lib = _lib.load()
    assert lib.vidu4d_surfel_geom_bytes(0) >= 256
    assert lib.vidu4d_surfel_geom_bytes(200000) > 200000 * 80
    assert lib.vidu4d_surfel_geom_bytes(200001) >= lib.vidu4d_surfel_geom_bytes(200000)
    assert lib.vidu4d_surfel_image_bytes(512, 512) >= 512 * 512 * 20 + 1024 * 8
    assert lib.vidu4d_surfel_binning_bytes(0) >= 256
    assert lib.vidu4d_surfel_binning_bytes(1 << 20) >= (1 << 20) * 20  # entries + scratch (u64) + point_list (u32)
    assert lib.vidu4d_surfel_backward_workspace_bytes(1000) >= 1000 * 80


def test_argument_validation_without_gpu():
    lib = _lib.load()
    a = _lib.ForwardArgs()
    a.P, a.width, a.height = 10, 0, 16  # bad width
    assert lib.vidu4d_surfel_forward_plan(C.byref(a), None) == -1
    assert b"bad sizes" in lib.vidu4d_last_error()
    a.width = 16
    a.transMat_precomp = 0x1000
    assert lib.vidu4d_surfel_forward_plan(C.byref(a), None) == -4  # upstream-undefined path is refused
    a.transMat_precomp = None
    a.tan_fovx = a.tan_fovy = 0.5
    a.shs, a.M, a.D = 0x1000, 4, 3  # degree 3 needs 16 coefficients
    assert lib.vidu4d_surfel_forward_plan(C.byref(a), None) == -1
    assert lib.vidu4d_quaternion_mul(4, None, 4, None, 4, None, None) == -1
    assert lib.vidu4d_quaternion_mul(4, 0x10, 5, 0x10, 4, 0x10, None) == -1
    with pytest.raises(_lib.SurfelError):
        _lib.check(-1, "x")


def test_boundary_module_surface():
    import inspect

    import diff_surfel_rasterization as d
    assert d.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    sig = inspect.signature(d.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    import torch
    r = d.GaussianRasterizer(None)
    x = torch.zeros(1, 3)
    with pytest.raises(Exception, match="excatly one of either SHs"):
        r(x, x, x)
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        r(x, x, x, shs=x)


def test_cpu_tensors_are_refused():
    import torch

    import diff_surfel_rasterization as d
    s = d.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                        torch.zeros(3), False, False)
    r = d.GaussianRasterizer(s)
    n = 4
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(torch.zeros(n, 3), torch.zeros(n, 3), torch.ones(n, 1), shs=torch.zeros(n, 1, 3), scales=torch.ones(n, 2),
          rotations=torch.ones(n, 4))
