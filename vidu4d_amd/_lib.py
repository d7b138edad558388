"""ctypes binding of libvidu4d_surfel.so (C ABI in include/vidu4d_surfel.h).

The product path has NO fallback: if the shared object is missing it is built with hipcc, and if
that fails (or the symbols are missing) importing a symbol raises.  Nothing here touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvidu4d_surfel.so")

ABI_VERSION = 21


class ForwardArgs(C.Structure):
    """struct Vidu4dSurfelForwardArgs"""
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int), ("debug", C.c_int),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("transMat_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("out_color", C.c_void_p), ("out_others", C.c_void_p), ("radii", C.c_void_p),
        ("geom_buffer", C.c_void_p), ("geom_bytes", C.c_size_t), ("image_buffer", C.c_void_p),
        ("image_bytes", C.c_size_t), ("segment_split", C.c_int), ("depth_used", C.c_void_p),
        ("frames", C.c_int), ("frame_viewmatrix", C.c_void_p * 8), ("frame_campos", C.c_void_p * 8),
        ("frame_tan_fovx", C.c_float * 8), ("frame_tan_fovy", C.c_float * 8),
        ("sh_dc", C.c_void_p), ("sh_rest", C.c_void_p), ("raw_params", C.c_int), ("aux_planes", C.c_int),
        ("assume_unsaturated", C.c_int), ("long_list_sort", C.c_int), ("debug_flags", C.c_int),
        ("max_list_hint", C.c_int),
    ]


class ContractJob(C.Structure):
    """struct Vidu4dContractJob"""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("rows_a", C.c_int), ("rows_b", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("sa", C.c_int64), ("sb", C.c_int64)]


class CopyJob(C.Structure):
    """struct Vidu4dCopyJob"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("src_ld", C.c_int64),
                ("src_cs", C.c_int64), ("dst_ld", C.c_int64)]


class AdamTensor(C.Structure):
    """struct Vidu4dAdamTensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64), ("lr", C.c_float), ("bias_correction1", C.c_float),
                ("bias_correction2_sqrt", C.c_float), ("device_scalars", C.c_void_p)]


class DensifyAttr(C.Structure):
    """struct Vidu4dDensifyAttr"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_m", C.c_void_p), ("dst_m", C.c_void_p),
                ("src_v", C.c_void_p), ("dst_v", C.c_void_p), ("width", C.c_int)]


class SkinFieldArgs(C.Structure):
    """struct Vidu4dSkinFieldArgs"""
    _fields_ = [("N", C.c_int), ("B", C.c_int), ("W", C.c_int), ("D", C.c_int)] + [
        (n, C.c_void_p) for n in ("xyz", "bone_A", "bone_c", "w_in", "b_in", "w_hid", "b_hid", "w_out", "b_out", "xbT",
                                  "rawT", "g_xbT", "g_rawT", "g_xyz", "relu_masks", "packed_fwd", "packed_bwd", "h_store",
                                  "g_store", "gx_store")] + [("h_store_rows", C.c_int)]


LOSS_MAX_FRAMES, LOSS_BLOCKS, LOSS_SUMS_FLOATS = 32, 1024, 32


class Stage3LossArgs(C.Structure):
    """struct Vidu4dStage3LossArgs"""
    _fields_ = [("M", C.c_int), ("H", C.c_int), ("W", C.c_int), ("color", C.c_void_p * LOSS_MAX_FRAMES),
                ("allmap", C.c_void_p * LOSS_MAX_FRAMES), ("bkgd", C.c_void_p), ("rgb", C.c_void_p),
                ("mask", C.c_void_p), ("vis2d", C.c_void_p), ("det", C.c_void_p), ("lambda_dssim", C.c_float),
                ("rgb_wt", C.c_float), ("mask_wt", C.c_float), ("dist_wt", C.c_float), ("sums", C.c_void_p),
                ("partials", C.c_void_p), ("losses", C.c_void_p), ("plane_stride", C.c_int64),
                ("normal_wt", C.c_float), ("depth_ratio", C.c_float), ("rays_d", C.c_void_p * LOSS_MAX_FRAMES),
                ("rays_o", C.c_void_p * LOSS_MAX_FRAMES), ("view3x3", C.c_void_p * LOSS_MAX_FRAMES),
                ("surf_normal", C.c_void_p * LOSS_MAX_FRAMES), ("surf_depth", C.c_void_p)]


class Stage3LossGrads(C.Structure):
    """struct Vidu4dStage3LossGrads"""
    _fields_ = [("g_color", C.c_void_p * LOSS_MAX_FRAMES), ("g_allmap", C.c_void_p * LOSS_MAX_FRAMES),
                ("g_bkgd", C.c_void_p), ("g_surf_normal", C.c_void_p * LOSS_MAX_FRAMES)]


SKIN_FIELD = dict(width=64, in_max=96, out_max=32, max_hidden=4)
AUX_ALPHA = 0x02  # VIDU4D_AUX_ALPHA
AUX_GEOM = 0x1F   # VIDU4D_AUX_GEOM: planes 0-4 (depth, alpha, normal)
DEBUG_NO_CULL, DEBUG_WHOLE_TILE_BACKWARD, DEBUG_SERIAL_REPAIR, DEBUG_POSITION_ORDER = 1, 2, 4, 8   # VIDU4D_DEBUG_*



def sched_pair(k: int) -> int:
    """VIDU4D_SCHED_PAIR(K) (ABI 21): two workgroups for the tiles longer than K / 4 x the mean list (15: every tile)"""
    return (int(k) & 15) << 12


def sched_xcd_block(block: int) -> int:
    """VIDU4D_SCHED_XCD_BLOCK(B): the schedule bits of Vidu4dSurfelForwardArgs::debug_flags"""
    return (int(block) & 15) << 8


BLEND_STATS = 11  # VIDU4D_BLEND_STATS (vidu4d_surfel_diag.h)
ADAM_MAX_TENSORS = 8
ADAMW_MAX_TENSORS = 32
CONTRACT_MAX_JOBS = 4
COPY_MAX_JOBS = 8
CLIP_MAX_TENSORS = 96
CLIP_WORKSPACE_FLOATS = 1056
DENSIFY_MAX_ATTRS = 8


class BackwardArgs(C.Structure):
    """struct Vidu4dSurfelBackwardArgs"""
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float), ("debug", C.c_int),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("radii", C.c_void_p), ("shs", C.c_void_p),
        ("colors_precomp", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
        ("transMat_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("dL_dout_color", C.c_void_p), ("dL_dout_others", C.c_void_p),
        ("geom_buffer", C.c_void_p), ("binning_buffer", C.c_void_p), ("binning_capacity", C.c_int64),
        ("image_buffer", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dL_dmeans2D", C.c_void_p), ("dL_dcolors", C.c_void_p), ("dL_dopacity", C.c_void_p),
        ("dL_dmeans3D", C.c_void_p), ("dL_dtransMat", C.c_void_p), ("dL_dsh", C.c_void_p),
        ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p), ("segment_split", C.c_int),
        ("frames", C.c_int), ("frame_viewmatrix", C.c_void_p * 8), ("frame_campos", C.c_void_p * 8),
        ("frame_tan_fovx", C.c_float * 8), ("frame_tan_fovy", C.c_float * 8),
        ("sh_dc", C.c_void_p), ("sh_rest", C.c_void_p), ("dL_dsh_dc", C.c_void_p), ("dL_dsh_rest", C.c_void_p),
        ("raw_params", C.c_int), ("aux_planes", C.c_int), ("debug_flags", C.c_int),
        ("diag_walk_counters", C.c_void_p),
    ]


class DenseStack(C.Structure):
    """Vidu4dDenseStack (include/vidu4d_surfel.h)."""
    MAX_LAYERS, MAX_WIDTH, MAX_ROWS = 16, 256, 16
    _fields_ = [("rows", C.c_int), ("n_trunk", C.c_int), ("n_head_a", C.c_int), ("n_head_b", C.c_int),
                ("in_", C.c_int * 16), ("out", C.c_int * 16), ("relu", C.c_int * 16), ("scale", C.c_float * 16),
                ("W", C.c_void_p * 16), ("b", C.c_void_p * 16), ("gW", C.c_void_p * 16), ("gb", C.c_void_p * 16)]


# every symbol include/vidu4d_surfel.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "vidu4d_surfel_abi_version": (C.c_int, []),
    "vidu4d_last_error": (C.c_char_p, []),
    "vidu4d_surfel_geom_bytes": (C.c_size_t, [C.c_int]),
    "vidu4d_surfel_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vidu4d_surfel_image_bytes_frames": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vidu4d_surfel_binning_bytes": (C.c_size_t, [C.c_int64]),
    "vidu4d_surfel_backward_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vidu4d_surfel_forward_plan": (C.c_int, [C.POINTER(ForwardArgs), _P]),
    "vidu4d_surfel_num_rendered": (C.c_int, [C.POINTER(ForwardArgs), _P, C.POINTER(C.c_int64)]),
    "vidu4d_surfel_forward_run": (C.c_int, [C.POINTER(ForwardArgs), _P, C.c_size_t, C.c_int64, _P]),
    "vidu4d_surfel_backward": (C.c_int, [C.POINTER(BackwardArgs), _P]),
    "vidu4d_surfel_mark_visible": (C.c_int, [C.c_int, _P, _P, _P, _P, _P]),
    "vidu4d_surfel_state_read": (C.c_int, [C.POINTER(ForwardArgs), _P, C.c_int64, C.c_int, _P, C.c_size_t,
                                           C.POINTER(C.c_int64), _P]),
    "vidu4d_diag_copy": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, _P]),
    "vidu4d_surfel_profile_enable": (C.c_int, [C.c_int]),
    "vidu4d_surfel_profile_stage_count": (C.c_int, []),
    "vidu4d_surfel_profile_stage_name": (C.c_char_p, [C.c_int]),
    "vidu4d_surfel_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "vidu4d_quaternion_mul": (C.c_int, [C.c_int64, _P, C.c_int, _P, C.c_int, _P, _P]),
    "vidu4d_quaternion_mul_backward": (C.c_int, [C.c_int64, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P]),
    "vidu4d_quaternion_mul_backward_backward": (C.c_int, [C.c_int64, _P, _P, _P, _P, C.c_int, _P, C.c_int, _P, _P,
                                                          _P, _P]),
    "vidu4d_quaternion_conjugate": (C.c_int, [C.c_int64, _P, _P, _P]),
    "vidu4d_lbs_forward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_P] * 10),
    "vidu4d_lbs_backward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_P] * 13),
    "vidu4d_lbs_skin_forward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_P] * 10 + [C.c_int, _P, _P, _P, C.c_int, _P]),
    "vidu4d_lbs_skin_backward": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_P] * 14 + [C.c_int, _P, _P, _P, C.c_int, _P, _P]),
    "vidu4d_lbs_skin_param_rows": (C.c_int, [C.c_int]),
    "vidu4d_bone_tables_forward": (C.c_int, [C.c_int, C.c_int] + [_P] * 10),
    "vidu4d_bone_tables_backward": (C.c_int, [C.c_int, C.c_int] + [_P] * 15),
    "vidu4d_camera_tail_forward": (C.c_int, [C.c_int, _P, _P, _P, _P]),
    "vidu4d_camera_tail_backward": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P]),
    "vidu4d_dense_stack_acts_floats": (C.c_int, [C.POINTER(DenseStack)]),
    "vidu4d_dense_stack_forward": (C.c_int, [C.POINTER(DenseStack), _P, _P, _P]),
    "vidu4d_dense_stack_backward": (C.c_int, [C.POINTER(DenseStack), _P, _P, _P, _P, _P, _P, _P]),
    "vidu4d_knn_mean_dist2": (C.c_int, [C.c_int, _P, _P, _P]),
    "vidu4d_radius_count": (C.c_int, [C.c_int, _P, C.c_float, _P, _P]),
    "vidu4d_post_forward": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P]),
    "vidu4d_post_backward": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "vidu4d_skin_field_forward": (C.c_int, [C.POINTER(SkinFieldArgs), _P]),
    "vidu4d_skin_field_backward": (C.c_int, [C.POINTER(SkinFieldArgs), _P]),
    "vidu4d_skin_field_packed_floats": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "vidu4d_skin_field_pack": (C.c_int, [C.POINTER(SkinFieldArgs), C.c_int, _P, _P]),
    "vidu4d_stage3_loss_forward": (C.c_int, [C.POINTER(Stage3LossArgs), _P]),
    "vidu4d_stage3_loss_backward": (C.c_int, [C.POINTER(Stage3LossArgs), _P, C.POINTER(Stage3LossGrads), _P]),
    "vidu4d_adam_step": (C.c_int, [C.c_int, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double, _P, C.c_int, _P]),
    "vidu4d_adam_step_guarded": (C.c_int, [C.c_int, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double, _P, C.c_int, _P, _P]),
    "vidu4d_adamw_step_guarded": (C.c_int, [C.c_int, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double, C.c_double, _P, C.c_int, _P, _P]),
    "vidu4d_contract_rows": (C.c_int, [C.c_int, C.POINTER(ContractJob), C.c_int64, _P]),
    "vidu4d_copy_strided": (C.c_int, [C.c_int, C.POINTER(CopyJob), _P]),
    "vidu4d_grad_clip_coef": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_float, _P, _P, _P]),
    "vidu4d_densify_plan": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "vidu4d_densify_apply": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DensifyAttr), C.c_int,
                                       C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
}

STATE = dict(num_rendered=0, records=1, tiles_touched=2, point_list=3, sorted_keys=4, ranges=5, final_T=6,
             n_contrib=7, tile_order=8, tail_order=9, header=10)

_lib = None


def load() -> C.CDLL:
    """Loads (building first if needed) the HIP library; raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles the HIP runtime (libamdhip64) that owns the device memory and
    # streams we are handed; loading ours first would bring a second runtime into the process.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    got = lib.vidu4d_surfel_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libvidu4d_surfel.so ABI {got} != expected {ABI_VERSION}: rebuild (python -m vidu4d_amd.build)")
    _lib = lib
    return lib


class SurfelError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        msg = load().vidu4d_last_error()
        raise SurfelError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def profile_enable(on: bool):
    load().vidu4d_surfel_profile_enable(int(on))


def profile_read(reset: bool = True) -> dict:
    """{stage name: (total milliseconds, launch-span count)} since the last reset."""
    lib = load()
    n = lib.vidu4d_surfel_profile_stage_count()
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    check(lib.vidu4d_surfel_profile_read(ms, cnt, int(reset)), "profile_read")
    return {lib.vidu4d_surfel_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
