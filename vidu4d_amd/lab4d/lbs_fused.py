"""Fused bob-LBS apply on the HIP path: skinning weights + per-frame bone dual quaternions + camera
-> camera-space surfel centres and orientations, one kernel forward and one backward (csrc/lbs.hip).

Equivalent to bob_warp.dual_quaternion_skinning_qt -> apply_qt_to_gaussian -> field2cam
apply_qt_to_gaussian (reference: lab4d/utils/geom_utils.py:48-92,
lab4d/nnutils/deformable_gaussian.py:1032-1046, :1425-1430), for bone / camera parameters that do not
require gradients (--gs_optim_warp=False)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from .. import _lib


def _c(t):
    return t.detach().float().contiguous()


class _LbsApply(Function):
    @staticmethod
    def forward(ctx, skin_prob, se3_qr, se3_qd, xyz, rot, cam_q, cam_t):
        if not xyz.is_cuda:
            raise RuntimeError("lbs_apply: HIP tensors required")
        for t, name in ((se3_qr, "se3"), (se3_qd, "se3"), (cam_q, "field2cam"), (cam_t, "field2cam")):
            if t.requires_grad:
                raise RuntimeError(f"lbs_apply: {name} requires grad; the fused path treats it as constant")
        M, B = se3_qr.shape[:2]
        N = xyz.shape[0]
        wT = _c(skin_prob).t().contiguous()  # (B,N): coalesced per-bone reads
        args = [wT, _c(se3_qr), _c(se3_qd), _c(xyz), _c(rot), _c(cam_q), _c(cam_t)]
        out_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=xyz.device)
        out_rot = torch.empty(M, N, 4, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        _lib.check(lib.vidu4d_lbs_forward(M, N, B, *[a.data_ptr() for a in args], out_xyz.data_ptr(),
                                          out_rot.data_ptr(), torch.cuda.current_stream(xyz.device).cuda_stream),
                   "lbs forward")
        ctx.save_for_backward(*args)
        ctx.dims = (M, N, B)
        return out_xyz, out_rot

    @staticmethod
    def backward(ctx, g_xyz_out, g_rot_out):
        M, N, B = ctx.dims
        args = ctx.saved_tensors
        dev = args[0].device
        g_xyz_out = torch.zeros(M, N, 3, device=dev) if g_xyz_out is None else _c(g_xyz_out)
        g_rot_out = torch.zeros(M, N, 4, device=dev) if g_rot_out is None else _c(g_rot_out)
        g_wT = torch.empty(M, B, N, dtype=torch.float32, device=dev)
        g_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=dev)
        g_rot = torch.empty(M, N, 4, dtype=torch.float32, device=dev)
        lib = _lib.load()
        _lib.check(lib.vidu4d_lbs_backward(M, N, B, *[a.data_ptr() for a in args], g_xyz_out.data_ptr(),
                                           g_rot_out.data_ptr(), g_wT.data_ptr(), g_xyz.data_ptr(), g_rot.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "lbs backward")
        return g_wT.sum(0).t(), None, None, g_xyz.sum(0), g_rot.sum(0), None, None


def lbs_apply(skin_prob, se3, xyz, rot, cam_q, cam_t):
    """skin_prob (N,B) softmax weights; se3 = (qr, qd) each (M,B,4); xyz (N,3); rot (N,4) raw
    orientations; cam_q (M,4), cam_t (M,3).  -> xyz_cam (M,N,3), rot_cam (M,N,4)."""
    return _LbsApply.apply(skin_prob, se3[0], se3[1], xyz, rot, cam_q, cam_t)


class _LbsSkinApply(Function):
    """xbT (3B,N) bone coordinates -- or None with the bone map (bone_A (3B,3), bone_c (3B)) of the rest pose: the kernels
    then evaluate x_bone = A xyz + c themselves --, rawT (B,N) raw delta-MLP output or None -> softmax skinning weights
    -> blend -> apply -> camera, for all frames, one kernel per direction (csrc/lbs.hip lbs_skin_kernel)."""

    @staticmethod
    def forward(ctx, xbT, rawT, se3_qr, se3_qd, xyz, rot, cam_q, cam_t, unit_rot=False, bone_A=None, bone_c=None,
                frame_index=None):
        if not xyz.is_cuda:
            raise RuntimeError("lbs_skin_apply: HIP tensors required")
        # bones and cameras that train (--gs_optim_warp=True, the reference's default): the backward reduces their gradients
        # over the surfels (csrc/lbs.hip, g_params).  That instance takes the bone coordinates as an input -- the bone MAP's
        # gradient flows through whatever made xbT -- and per-frame rows.
        trainable = any(t.requires_grad for t in (se3_qr, se3_qd, cam_q, cam_t))
        if bone_A is not None and (bone_A.requires_grad or bone_c.requires_grad):
            raise RuntimeError("lbs_skin_apply: a bone map that requires grad must come in as xbT = A xyz + c (torch)")
        if trainable and (bone_A is not None or frame_index is not None):
            raise RuntimeError("lbs_skin_apply: bones / cameras that require grad need xbT (not the in-kernel bone map) and "
                               "per-frame rows (no frame_index)")
        M, B = se3_qr.shape[:2]
        if frame_index is not None:   # se3 / cam are tables over all frames; this call's frames are their rows frame_index
            if frame_index.dtype != torch.int64 or not frame_index.is_cuda or frame_index.ndim != 1:
                raise RuntimeError("lbs_skin_apply: frame_index must be a 1-d int64 tensor on the device")
            M = frame_index.shape[0]
            frame_index = frame_index.contiguous()
        table_rows = int(se3_qr.shape[0])
        N = xyz.shape[0]
        if (xbT is None) == (bone_A is None):
            raise RuntimeError("lbs_skin_apply: pass the bone coordinates xbT OR the bone map (bone_A, bone_c)")
        if xbT is not None and xbT.shape != (3 * B, N) or (rawT is not None and rawT.shape != (B, N)):
            raise RuntimeError(f"lbs_skin_apply: xbT / rawT must be (3B, N) / (B, N) with B={B}, N={N}")
        if bone_A is not None and (bone_A.shape != (3 * B, 3) or bone_c.shape != (3 * B,)):
            raise RuntimeError(f"lbs_skin_apply: bone map must be (3B, 3) / (3B,) with B={B}")
        args = [None if xbT is None else _c(xbT), None if rawT is None else _c(rawT), _c(se3_qr), _c(se3_qd), _c(xyz),
                _c(rot), _c(cam_q), _c(cam_t)]
        bmap = [None, None] if bone_A is None else [_c(bone_A), _c(bone_c)]
        out_xyz = torch.empty(M, N, 3, dtype=torch.float32, device=xyz.device)
        out_rot = torch.empty(M, N, 4, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        ptr = [None if a is None else a.data_ptr() for a in args]
        bptr = [None if a is None else a.data_ptr() for a in bmap]
        _lib.check(lib.vidu4d_lbs_skin_forward(M, N, B, *ptr, out_xyz.data_ptr(), out_rot.data_ptr(), int(unit_rot), *bptr,
                                               None if frame_index is None else frame_index.data_ptr(), table_rows,
                                               torch.cuda.current_stream(xyz.device).cuda_stream), "lbs skin forward")
        ctx.frame_index, ctx.table_rows, ctx.trainable = frame_index, table_rows, trainable
        ctx.present = [a is not None for a in args + bmap]
        ctx.unit_rot = bool(unit_rot)
        ctx.save_for_backward(*[a for a in args + bmap if a is not None])
        ctx.dims = (M, N, B)
        return out_xyz, out_rot

    @staticmethod
    def backward(ctx, g_xyz_out, g_rot_out):
        M, N, B = ctx.dims
        it = iter(ctx.saved_tensors)
        full = [next(it) if have else None for have in ctx.present]
        saved, bmap = full[:8], full[8:]
        has_xb, has_raw = saved[0] is not None, saved[1] is not None
        dev = saved[4].device
        g_xyz_out = torch.zeros(M, N, 3, device=dev) if g_xyz_out is None else _c(g_xyz_out)
        g_rot_out = torch.zeros(M, N, 4, device=dev) if g_rot_out is None else _c(g_rot_out)
        g_xbT = torch.empty(3 * B, N, dtype=torch.float32, device=dev) if has_xb else None
        g_rawT = torch.empty(B, N, dtype=torch.float32, device=dev) if has_raw else None
        g_xyz = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g_rot = torch.empty(N, 4, dtype=torch.float32, device=dev)
        lib = _lib.load()
        ptr = [None if a is None else a.data_ptr() for a in saved]
        bptr = [None if a is None else a.data_ptr() for a in bmap]
        g_params = None
        if ctx.trainable:   # one row of partial sums per workgroup of 256 surfels: (rows, M, 8 B + 8)
            g_params = torch.empty(lib.vidu4d_lbs_skin_param_rows(N), M, 8 * B + 8, dtype=torch.float32, device=dev)
        _lib.check(lib.vidu4d_lbs_skin_backward(M, N, B, *ptr, g_xyz_out.data_ptr(), g_rot_out.data_ptr(),
                                                None if g_xbT is None else g_xbT.data_ptr(),
                                                None if g_rawT is None else g_rawT.data_ptr(), g_xyz.data_ptr(),
                                                g_rot.data_ptr(), int(ctx.unit_rot), *bptr,
                                                None if ctx.frame_index is None else ctx.frame_index.data_ptr(),
                                                ctx.table_rows, None if g_params is None else g_params.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream),
                   "lbs skin backward")
        g_qr = g_qd = g_cq = g_ct = None
        if g_params is not None:
            tot = g_params.sum(0)                                   # (M, 8 B + 8)
            bones = tot[:, :8 * B].view(M, B, 8)
            need = ctx.needs_input_grad
            g_qr = bones[..., :4].contiguous() if need[2] else None
            g_qd = bones[..., 4:].contiguous() if need[3] else None
            g_cq = tot[:, 8 * B:8 * B + 4].contiguous() if need[6] else None
            g_ct = tot[:, 8 * B + 4:8 * B + 7].contiguous() if need[7] else None
        return g_xbT, g_rawT, g_qr, g_qd, g_xyz, g_rot, g_cq, g_ct, None, None, None, None


LBS_MAX_FRAMES = 8   # (csrc/lbs.hip MAX_FRAMES: frames of one launch)


def lbs_skin_apply(xbT, rawT, se3, xyz, rot, cam_q, cam_t, unit_rot=False, bone_map=None, frame_index=None):
    """xbT (3B,N) Gaussian-bone coordinates; rawT (B,N) raw output of the delta-skin MLP or None; se3 = (qr, qd)
    each (M,B,4) (more than 8 frames: groups of 8 launches); xyz (N,3); rot (N,4); cam_q (M,4), cam_t (M,3).  -> xyz_cam (M,N,3), rot_cam (M,N,4);
    unit_rot: rot_cam comes out normalised (F.normalize, the renderer's rotation activation, fused in).
    bone_map = (A (3B,3), c (3B)) with xbT = None (frozen bones): the kernels evaluate x_bone = A xyz + c themselves and
    the gradient w.r.t. xyz includes that path -- the (3B,N) coordinates and their gradient never cross HBM."""
    bone_A, bone_c = (None, None) if bone_map is None else bone_map
    M = cam_q.shape[0] if frame_index is None else frame_index.shape[0]
    if M <= LBS_MAX_FRAMES:
        return _LbsSkinApply.apply(xbT, rawT, se3[0], se3[1], xyz, rot, cam_q, cam_t, unit_rot, bone_A, bone_c, frame_index)
    # More frames than one launch takes (the kernels keep a step's bone tables in LDS: 8 frames): groups of <= 8, their
    # outputs concatenated; autograd adds the groups' gradients of everything the frames share.  The reference's loop takes
    # any imgs_per_gpu (lab4d/nnutils/deformable_gaussian.py:1175).
    xs, rs = [], []
    for lo in range(0, M, LBS_MAX_FRAMES):
        sl = slice(lo, lo + LBS_MAX_FRAMES)
        if frame_index is None:
            x, r = _LbsSkinApply.apply(xbT, rawT, se3[0][sl], se3[1][sl], xyz, rot, cam_q[sl], cam_t[sl], unit_rot, bone_A,
                                       bone_c, None)
        else:   # (se3 / cam_* are tables over all frames: the group's frame ids index them)
            x, r = _LbsSkinApply.apply(xbT, rawT, se3[0], se3[1], xyz, rot, cam_q, cam_t, unit_rot, bone_A, bone_c,
                                       frame_index[sl].contiguous())
        xs.append(x)
        rs.append(r)
    return torch.cat(xs, 0), torch.cat(rs, 0)


# ---------------------------------------------------------------------------------------------------------------
# Bone coordinates + delta-skin MLP in one kernel per direction (csrc/skin_field.hip)
def skin_field_supported(sm) -> bool:
    """True when SkinningField `sm` has the shape csrc/skin_field.hip is written for (the bob field: width 64,
    no positional encoding of the bone coordinates, no skip connection inside the stack) and frozen weights."""
    return _skin_field_frozen(sm)


def skin_field_shape_supported(sm) -> bool:
    """The shape condition alone (skin_field_train serves weights that require grad)."""
    lim = _lib.SKIN_FIELD
    if not sm.has_delta or sm.num_freq_xyz != 0:
        return False
    mlp = sm.delta_field
    B = sm.log_gauss.shape[0]
    if mlp.W != lim["width"] or not (1 <= mlp.D <= lim["max_hidden"]) or any(0 < s_ < mlp.D for s_ in mlp.skips):
        return False
    if 3 * B > lim["in_max"] or B > lim["out_max"] or sm.xyz_channels != 3 * B:
        return False
    return True


def _skin_field_frozen(sm) -> bool:
    if not skin_field_shape_supported(sm):
        return False
    plist = sm.__dict__.get("_param_list")   # (collected once: Module.parameters() walks the module tree, every step)
    if plist is None:
        plist = sm.__dict__["_param_list"] = list(sm.parameters())
    return not any(p.requires_grad for p in plist)


def prepare_skin_field(sm, A, c0) -> dict:
    """Device arrays in the layout Vidu4dSkinFieldArgs documents (row-major, zero-padded), from the SkinningField's
    weights and the rest pose's bone map x_bone = A xyz + c0."""
    lim = _lib.SKIN_FIELD
    W, IN, OUT = lim["width"], lim["in_max"], lim["out_max"]
    mlp = sm.delta_field
    dev = A.device
    B3 = A.shape[0]
    with torch.no_grad():
        def pad(t, shape):
            out = torch.zeros(shape, dtype=torch.float32, device=dev)
            out[tuple(slice(0, n) for n in t.shape)] = t.float()
            return out

        w1 = mlp.linear_1[0].weight[:, :B3]                       # (W, 3B): the coordinate columns
        hid = [getattr(mlp, f"linear_{i + 1}")[0] for i in range(1, mlp.D)]
        wo, bo = mlp.linear_final.weight, mlp.linear_final.bias    # (B, W), (B)
        tab = {"B": wo.shape[0], "D": mlp.D, "bone_A": A.float().contiguous(), "bone_c": c0.float().contiguous(),
               "w_in": pad(w1, (W, IN)), "w_out": pad(wo, (OUT, W)), "b_out": pad(bo, (OUT,))}
        if hid:
            tab["w_hid"] = torch.stack([l_.weight for l_ in hid]).float().contiguous()
            tab["b_hid"] = torch.stack([l_.bias for l_ in hid]).float().contiguous()
    return tab


def pack_skin_field(tab) -> None:
    """Adds tab["packed_fwd"] / tab["packed_bwd"]: the weights as the kernels keep them in LDS (vidu4d_skin_field_pack),
    so that a launch copies 46 / 75 KB linearly instead of gathering them (~20 us per launch).  The tables are constants
    of a fused-path model (skin_field_supported: frozen weights); call again after changing them."""
    dev = tab["bone_A"].device
    lib = _lib.load()
    a = _lib.SkinFieldArgs()
    a.N, a.B, a.W, a.D = 0, tab["B"], _lib.SKIN_FIELD["width"], tab["D"]
    for k in ("bone_A", "bone_c", "w_in", "w_hid", "b_hid", "w_out", "b_out"):
        setattr(a, k, None if tab.get(k) is None else tab[k].data_ptr())
    st = torch.cuda.current_stream(dev).cuda_stream
    for backward, key in ((0, "packed_fwd"), (1, "packed_bwd")):
        n = lib.vidu4d_skin_field_packed_floats(tab["B"], tab["D"], backward)
        if n <= 0:
            raise RuntimeError("skin_field: unsupported shape")
        out = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(lib.vidu4d_skin_field_pack(a, backward, out.data_ptr(), st), "skin field pack")
        tab[key] = out


def _skin_field_args(tab, N, xyz, b_in, **ptrs):
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    a = _lib.SkinFieldArgs()
    a.N, a.B, a.W, a.D = N, tab["B"], _lib.SKIN_FIELD["width"], tab["D"]
    a.xyz, a.b_in = xyz.data_ptr(), b_in.data_ptr()
    for k in ("bone_A", "bone_c", "w_in", "w_hid", "b_hid", "w_out", "b_out", "packed_fwd", "packed_bwd"):
        setattr(a, k, p(tab.get(k)))
    for k, v in ptrs.items():
        setattr(a, k, p(v))
    return a


class _SkinField(Function):
    @staticmethod
    def forward(ctx, xyz, b_in, tab, want_xb=True):
        if not xyz.is_cuda:
            raise RuntimeError("skin_field: HIP tensors required")
        if b_in.requires_grad:
            raise RuntimeError("skin_field: the first-layer bias (time / instance code) requires grad; the fused path "
                               "treats the skinning network as constant")
        N = xyz.shape[0]
        x, b = _c(xyz), _c(b_in).reshape(-1)
        if "packed_fwd" not in tab and tab.get("pack", True):
            pack_skin_field(tab)
        xbT = torch.empty(3 * tab["B"], N, dtype=torch.float32, device=xyz.device) if want_xb else None
        rawT = torch.empty(tab["B"], N, dtype=torch.float32, device=xyz.device)
        # which hidden units are active, per layer / 32-surfel tile / lane: spares the backward its recomputation
        masks = torch.empty(tab["D"] * 64 * ((N + 31) // 32), dtype=torch.int32, device=xyz.device)
        a = _skin_field_args(tab, N, x, b, xbT=xbT, rawT=rawT, relu_masks=masks)
        _lib.check(_lib.load().vidu4d_skin_field_forward(a, torch.cuda.current_stream(xyz.device).cuda_stream),
                   "skin field forward")
        ctx.save_for_backward(x, b, masks)
        ctx.tab = tab
        return xbT, rawT

    @staticmethod
    def backward(ctx, g_xbT, g_rawT):
        x, b, masks = ctx.saved_tensors
        tab, N = ctx.tab, x.shape[0]
        g_xbT = None if g_xbT is None else _c(g_xbT)
        g_rawT = torch.zeros(tab["B"], N, device=x.device) if g_rawT is None else _c(g_rawT)
        g_xyz = torch.empty(N, 3, dtype=torch.float32, device=x.device)
        a = _skin_field_args(tab, N, x, b, g_xbT=g_xbT, g_rawT=g_rawT, g_xyz=g_xyz, relu_masks=masks)
        _lib.check(_lib.load().vidu4d_skin_field_backward(a, torch.cuda.current_stream(x.device).cuda_stream),
                   "skin field backward")
        return g_xyz, None, None, None


def skin_field(xyz, b_in, tab, want_xb=True):
    """xyz (N,3) canonical centres; b_in (W,) first-layer bias of the step (SkinningField.frame_bias); tab from
    prepare_skin_field.  -> xbT (3B,N) Gaussian-bone coordinates (None when want_xb is False: lbs_skin_apply with
    bone_map evaluates them itself), rawT (B,N) raw delta-skin output."""
    return _SkinField.apply(xyz, b_in, tab, want_xb)


# ---- the same kernels for networks that TRAIN (--gs_optim_warp=True): TRAIN instances leave the hidden activations and the
# masked pre-activation gradients in feature-major arrays; the weight gradients are contractions over the surfels
# (bob_warp.contract_over_columns: batched GEMMs over K-chunks), the bias gradients row sums.
def _train_buffers(sm, dev, N, B, scratch=False):
    """Persistent arrays of the TRAIN path: the padded weights in the kernels' layout (padding stays zero; one copy each per
    step) and, per surfel count, what the contractions over the surfels read -- the bone coordinates, every hidden layer's
    activations and the homogeneous centres, each block FOLLOWED BY A ROW OF ONES (written once): g [x; 1]^T gives the weight
    gradient and the bias gradient in one batched GEMM instead of a GEMM and a 50 MB row sum."""
    lim = _lib.SKIN_FIELD
    W, IN, OUT = lim["width"], lim["in_max"], lim["out_max"]
    bufs = sm.__dict__.get("_skin_field_train_bufs")
    D = sm.delta_field.D
    if bufs is None or bufs["w_in"].device != dev:
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
        bufs = sm.__dict__["_skin_field_train_bufs"] = {"w_in": z(W, IN), "w_out": z(OUT, W), "b_out": z(OUT),
                                                        "w_hid": z(max(D - 1, 1), W, W), "b_hid": z(max(D - 1, 1), W)}
    if scratch:
        # (a forward nobody differentiates -- torch.no_grad() -- must not touch the per-surfel arrays a pending backward of
        # this model still reads: temporaries, sized as the persistent ones; the padded weights are rewritten with the same
        # values either way)
        return dict(bufs, N=N, gen=None, xb1=torch.empty(3 * B + 1, N, dtype=torch.float32, device=dev),
                    h1=torch.empty(D, W + 1, N, dtype=torch.float32, device=dev),
                    xyz1=torch.empty(4, N, dtype=torch.float32, device=dev))
    if bufs.get("N") != N:
        bufs["N"] = N
        xb1 = torch.empty(3 * B + 1, N, dtype=torch.float32, device=dev)
        xb1[3 * B] = 1.0
        h1 = torch.empty(D, W + 1, N, dtype=torch.float32, device=dev)
        h1[:, W] = 1.0
        xyz1 = torch.ones(4, N, dtype=torch.float32, device=dev)   # ([x; y; z; 1] feature-major, like the other right operands)
        bufs.update(xb1=xb1, h1=h1, xyz1=xyz1)
    bufs["gen"] = bufs.get("gen") or 0
    return bufs


def copy_strided(copies):
    """[(src (R, C) view, dst (R, C) view with unit column stride), ...] -> dst[...] = src, one launch per eight copies
    (csrc/contract.hip copy_strided_kernel); fp32 HIP tensors."""
    lib = _lib.load()
    dev = copies[0][1].device
    stream = torch.cuda.current_stream(dev).cuda_stream
    for s0 in range(0, len(copies), _lib.COPY_MAX_JOBS):
        jobs = []
        for src, dst in copies[s0:s0 + _lib.COPY_MAX_JOBS]:
            if src.shape != dst.shape or src.dim() != 2 or src.dtype != torch.float32 or dst.dtype != torch.float32 or \
                    not dst.is_cuda or (dst.shape[1] > 1 and dst.stride(1) != 1):
                raise RuntimeError("copy_strided: 2-D fp32 HIP views of equal shape, unit column stride on the destination")
            jobs.append(_lib.CopyJob(src.data_ptr(), dst.data_ptr(), int(src.shape[0]), int(src.shape[1]), int(src.stride(0)),
                                     int(src.stride(1)), int(dst.stride(0))))
        arr = (_lib.CopyJob * len(jobs))(*jobs)
        _lib.check(lib.vidu4d_copy_strided(len(jobs), arr, stream), "copy strided")


def contract_pairs(pairs):
    """[(L (O, N), R (I, N)), ...] -> [L @ R^T (O, I), ...]: the contractions over the surfels of a step's weight gradients in
    one launch per four of them (csrc/contract.hip; bob_warp.contract_over_columns is the library form: four launches
    each).  Operands: fp32 HIP tensors whose rows are strided views (unit or constant stride along N), at most 96 rows."""
    dev = pairs[0][0].device
    sizes = [(int(l.shape[0]), int(r.shape[0])) for l, r in pairs]
    flat = torch.zeros(sum(o * i for o, i in sizes), dtype=torch.float32, device=dev)   # (the kernel ADDS: one fill for all)
    outs, off = [], 0
    for o, i in sizes:
        outs.append(flat[off:off + o * i].view(o, i))
        off += o * i
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    N = int(pairs[0][0].shape[1])
    for s0 in range(0, len(pairs), _lib.CONTRACT_MAX_JOBS):
        chunk = list(zip(pairs, outs))[s0:s0 + _lib.CONTRACT_MAX_JOBS]
        jobs = []
        for (l, r), out in chunk:
            if l.dtype != torch.float32 or r.dtype != torch.float32 or not l.is_cuda or int(l.shape[1]) != N or int(r.shape[1]) != N \
                    or max(l.shape[0], r.shape[0]) > 96:
                raise RuntimeError("contract_pairs: fp32 HIP operands over the same N, at most 96 rows each")
            jobs.append(_lib.ContractJob(l.data_ptr(), r.data_ptr(), out.data_ptr(), int(l.shape[0]), int(r.shape[0]),
                                         int(l.stride(0)), int(r.stride(0)), int(l.stride(1)), int(r.stride(1))))
        arr = (_lib.ContractJob * len(jobs))(*jobs)
        _lib.check(lib.vidu4d_contract_rows(len(jobs), arr, N, stream), "contract rows")
    return outs


class _SkinFieldTrain(Function):
    @staticmethod
    def forward(ctx, xyz, b_in, A, c0, w1, w_out, b_out, sm_bufs, *hidden):
        """hidden = (w_2, b_2, ..., w_D, b_D); w1 (W, 3B) the first layer's coordinate columns; sm_bufs: _train_buffers."""
        if not xyz.is_cuda:
            raise RuntimeError("skin_field: HIP tensors required")
        N, B, D = xyz.shape[0], w_out.shape[0], 1 + len(hidden) // 2
        W = _lib.SKIN_FIELD["width"]
        x, b = _c(xyz), _c(b_in).reshape(-1)
        with torch.no_grad():
            # the weights into the kernels' padded arrays, the centres into the feature-major [x; y; z; 1]: one launch
            copies = [(w1, sm_bufs["w_in"][:, :3 * B]), (w_out, sm_bufs["w_out"][:B]), (b_out[None], sm_bufs["b_out"][None, :B]),
                      (x.t(), sm_bufs["xyz1"][:3])]
            for i in range(D - 1):
                copies += [(hidden[2 * i], sm_bufs["w_hid"][i]), (hidden[2 * i + 1][None], sm_bufs["b_hid"][i][None])]
            copy_strided(copies)
        tab = dict(B=B, D=D, bone_A=_c(A), bone_c=_c(c0), **{k: sm_bufs[k] for k in ("w_in", "w_out", "b_out", "w_hid", "b_hid")})
        dev = xyz.device
        xb1, h1 = sm_bufs["xb1"], sm_bufs["h1"]
        xbT = xb1[:3 * B]
        rawT = torch.empty(B, N, dtype=torch.float32, device=dev)
        masks = torch.empty(D * 64 * ((N + 31) // 32), dtype=torch.int32, device=dev)
        a = _skin_field_args(tab, N, x, b, xbT=xbT, rawT=rawT, relu_masks=masks, h_store=h1)
        a.h_store_rows = W + 1
        _lib.check(_lib.load().vidu4d_skin_field_forward(a, torch.cuda.current_stream(dev).cuda_stream), "skin field forward")
        ctx.save_for_backward(x, b, masks)
        ctx.tab, ctx.bufs = tab, sm_bufs   # (persistent arrays: unchanged until the next forward of this model)
        # ... which is checked, not assumed: xbT is a VIEW of the persistent xb1 and the backward reads xb1 / h1 /
        # xyz1 through raw pointers, so a second grad-enabled forward of the model before this one's backward would hand
        # autograd the wrong activations without any version counter noticing
        if sm_bufs.get("gen") is not None:
            sm_bufs["gen"] += 1
        ctx.gen = sm_bufs.get("gen")
        ctx.dims = (N, B, D, W)
        return xbT, rawT

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_xbT, g_rawT):
        from .bob_warp import contract_over_columns
        x, b, masks = ctx.saved_tensors
        N, B, D, W = ctx.dims
        dev = x.device
        if ctx.gen != ctx.bufs.get("gen"):
            raise RuntimeError("skin_field_train: the model's skinning field was evaluated again with autograd on between this "
                               "forward and its backward; the activations this backward reads live in per-model persistent "
                               "arrays and have been overwritten.  One grad-enabled forward per backward (renders under "
                               "torch.no_grad() in between are fine), or set fused_skin_field_trainable=False.")
        xb1, h1, xyz1 = ctx.bufs["xb1"], ctx.bufs["h1"], ctx.bufs["xyz1"]
        g_xbT = None if g_xbT is None else _c(g_xbT)
        g_rawT = torch.zeros(B, N, device=dev) if g_rawT is None else _c(g_rawT)
        g_xyz = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g = torch.empty(D, W, N, dtype=torch.float32, device=dev)
        gx = torch.empty(3 * B, N, dtype=torch.float32, device=dev)
        a = _skin_field_args(ctx.tab, N, x, b, g_xbT=g_xbT, g_rawT=g_rawT, g_xyz=g_xyz, relu_masks=masks, g_store=g, gx_store=gx)
        _lib.check(_lib.load().vidu4d_skin_field_backward(a, torch.cuda.current_stream(dev).cuda_stream), "skin field backward")
        need = ctx.needs_input_grad
        # every contraction against [x; 1]: the last column is the bias gradient (the row sum of its left operand)
        g_A = g_c = g_w1 = g_b_in = g_wo = g_bo = None
        pairs = []   # (left (O, N), right (I, N), what)
        if need[2] or need[3]:
            pairs.append((gx, xyz1, "A"))                      # (3B, 4)
        if need[4] or need[1]:
            pairs.append((g[0], xb1, "w1"))                    # (W, 3B + 1)
        if need[5] or need[6]:
            pairs.append((g_rawT, h1[D - 1], "wo"))            # (B, W + 1)
        for i in range(1, D):
            pairs.append((g[i], h1[i - 1], i))                 # (W, W + 1)
        outs = contract_pairs([(l, r) for l, r, _ in pairs]) if ctx.bufs.get("fused_contractions", True) else \
            [contract_over_columns(l, r) for l, r, _ in pairs]
        hidden = []
        for (_l, _r, what), t in zip(pairs, outs):
            if what == "A":
                g_A, g_c = t[:, :3], t[:, 3]
            elif what == "w1":
                g_w1, g_b_in = t[:, :3 * B], t[:, 3 * B]
            elif what == "wo":
                g_wo, g_bo = t[:, :W], t[:, W]
            else:
                hidden += [t[:, :W], t[:, W]]
        return (g_xyz if need[0] else None, g_b_in, g_A, g_c, g_w1, g_wo, g_bo, None, *hidden)


def skin_field_train(xyz, b_in, A, c0, sm):
    """skin_field for a SkinningField whose weights (and bone map, and time-code bias) require grad: xbT (3B,N), rawT (B,N)
    with gradients w.r.t. xyz, b_in, A, c0 and every weight and bias of the delta MLP."""
    mlp = sm.delta_field
    B3 = A.shape[0]
    hidden = []
    for i in range(1, mlp.D):
        lin = getattr(mlp, f"linear_{i + 1}")[0]
        hidden += [lin.weight, lin.bias]
    return _SkinFieldTrain.apply(xyz, b_in, A, c0, mlp.linear_1[0].weight[:, :B3], mlp.linear_final.weight,
                                 mlp.linear_final.bias, _train_buffers(sm, xyz.device, xyz.shape[0], mlp.linear_final.weight.shape[0],
                                                scratch=not torch.is_grad_enabled()),
                                 *hidden)
