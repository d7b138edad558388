"""Drop-in for the reference's native module `diff_surfel_rasterization._C`.

Same three functions, same positional arguments and return tuples as the pybind module
(/root/reference/gs/submodules/diff-surfel-rasterization/ext.cpp:15-19, rasterize_points.cu:39-60,
:143-166, :242-261); the body marshals torch tensors to raw device pointers and calls the C ABI
(include/vidu4d_surfel.h) on torch's current HIP stream.  There is no CPU path: tensors must live
on a GPU and the HIP library must load.

Difference that a caller can observe only through timing: `rasterize_gaussians` does not stall the
GPU in the middle of the forward to read `num_rendered`.  The binning buffer is sized from the
previous call's count (+25 %), all kernels are queued, and the host waits only for the tile-count
scan; if the guess was too small the tail of the forward is queued again with an exact buffer.
Set VIDU4D_SURFEL_EXACT=1 to size exactly first (one host sync before the sort, as upstream).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import threading

import torch

from . import _lib

_EXACT = os.environ.get("VIDU4D_SURFEL_EXACT", "0") == "1"
# Segment-parallel blending of long tile lists (csrc/blend.hip): "auto" switches it on for a frame
# when, in an earlier frame of the same shape, some pixel blended deeper than SPLIT_AUTO_LEN list
# entries (the serial chain of that tile then bounds the blend kernels; long lists that saturate
# early do not count -- splitting them only adds work); "1" / "0" force it on / off.
# Round 4: ... AND the frame had fewer long tiles than SPLIT_AUTO_TILES_PER_CU per compute unit.  Since the backward of a
# whole-tile forward runs segment-parallel anyway (recorded segments), the only thing the split still buys is parallelism
# for the FORWARD, and a frame with enough long tiles to fill the chip has it already -- while the segment-parallel forward
# blends every segment up to the caller's limit, the whole-tile walk stops where the pixels saturate.  Stage-3 ball of
# 200 k surfels at 512^2, two frames (round 4): radius 1.0 = 720 long tiles, depth 4.5 k: 1.18 ms whole / 1.23
# split (1.42 / 1.55 in the regularised regime); radius 0.7 = 350 tiles, 6.7 k: 1.30 / 1.19; 0.5: 1.59 / 1.21; 0.3: 2.78 / 2.00.
# Round 5: "long" is now the same count in either mode -- tiles longer than SPLIT_MIN = 1024 entries, Header word 14; the
# round-4 counts above were tiles above the recorded segments' 320-entry class after a whole-tile frame and positions up to
# the last 1024-entry tile after a split one, so the rule could stay in whichever mode ran first.  Same scene, same threshold
# (profiles/r05_split_rule.txt): radius 1.0 = 604 tiles: 1.27 whole / 1.24 split; 0.85 = 440: 1.25 / 1.14;
# 0.7 = 320: 1.25 / 1.05; 0.5 = 176: 1.52 / 1.02.  On the new count the threshold is 2.0 tiles per compute unit (512): radius
# 0.85 and below split, radius 1.0 and bench.py's fitting scene (a filled unit ball: whole 0.98-1.01 ms / split 1.00-1.13,
# three runs each, profiles/r05_split_rule.txt) walk whole tiles -- what round 4's rule did with its 2.5 on the other count.
# Round 6: the whole-tile forward gives its longest tiles two workgroups each (PAIR_K below), which takes most of its tail away:
# same scene (tools/experiments/pair_vs_split.sh, ms per step before / after step 8000): radius 1.0 whole + pairs 0.95 / 1.16
# against split 1.00 / 1.26; 0.85: 0.95 / 1.15 against 1.00-1.04 / 1.23-1.26; 0.7: 0.98 / 1.17 against 0.97-1.01 / 1.16-1.18;
# 0.5: 1.08 / 1.27 against 0.87-0.89 / 1.04-1.11.  The threshold moves to 1.25 tiles per compute unit (320): radius 0.85 now
# walks whole tiles.  (Without pairs -- VIDU4D_SURFEL_PAIR=0 -- it is round 5's 2.0.)
_SPLIT = os.environ.get("VIDU4D_SURFEL_SPLIT", "auto")
SPLIT_AUTO_LEN = int(os.environ.get("VIDU4D_SURFEL_SPLIT_AUTO_LEN", "2048"))
SPLIT_AUTO_TILES_PER_CU = float(os.environ.get("VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU",
                                               "1.25" if int(os.environ.get("VIDU4D_SURFEL_PAIR", "6")) > 0 else "2.0"))
SEG_LEN = int(os.environ.get("VIDU4D_SEG_LEN", "512"))   # csrc/surfel_state.h SEG_LEN (a variant build with another length says so here)
MSD_SORT_FROM = int(os.environ.get("VIDU4D_MSD_SORT_FROM", "10000"))  # longest list from which the long lists are MSD-split
# The segment-parallel alpha-only blend runs without its transmittance pre-pass (Vidu4dSurfelForwardArgs::
# assume_unsaturated): segments are blended from T = 1 and scaled in the combine, which blends the one segment a pixel
# saturates in again from the exact start (round 3; until then a saturating frame raised Header::truncated and was
# replayed with the pre-pass, so only callers that could replay speculated, and only on frames known not to saturate).
_SPEC = os.environ.get("VIDU4D_SURFEL_SPEC", "1") == "1"
# The same for the colour + planes-0-4 blend (AUX_GEOM): supported and tested, but OFF by default -- on the dense Stage-3
# ball (200 k surfels, lists of ~5 k entries, every covered pixel saturates) the combine pass's serial repair of the
# saturating segments costs more than the transmittance pre-pass it saves (blend_fwd + combine 250 + 311 us against
# seg_T + blend_fwd + combine 148 + 255 + 50, profiles/r04_fit_step_geometry_kernel_stats*.csv).
_SPEC_GEOM = os.environ.get("VIDU4D_SURFEL_SPEC_GEOM", "0") == "1"
# Debugging switches (Vidu4dSurfel*Args::debug_flags).  VIDU4D_SURFEL_NO_CULL=1: the blend kernels' footprint culls are off
# (every list entry is evaluated for every pixel of its tile, as the reference does); VIDU4D_SURFEL_WHOLE_TILE_BWD=1: the
# backward of an unsplit forward walks every tile with one workgroup (no recorded segments).  The process-wide default; a
# context (below) may override it for the calls made under it.
DEBUG_FLAGS = ((_lib.DEBUG_NO_CULL if os.environ.get("VIDU4D_SURFEL_NO_CULL", "0") == "1" else 0) |
               (_lib.DEBUG_WHOLE_TILE_BACKWARD if os.environ.get("VIDU4D_SURFEL_WHOLE_TILE_BWD", "0") == "1" else 0) |
               (_lib.DEBUG_SERIAL_REPAIR if os.environ.get("VIDU4D_SURFEL_SERIAL_REPAIR", "0") == "1" else 0) |
               (_lib.DEBUG_POSITION_ORDER if os.environ.get("VIDU4D_SURFEL_POSITION_ORDER", "0") == "1" else 0))
# The blend / sort launches' schedule (round 6): B > 0 = eight longest-first queues, one per XCD, the tiles of a BxB-tile block
# of a frame on one XCD (VIDU4D_SCHED_XCD_BLOCK, csrc/binning.hip grouped_order); 0 = one queue over all tiles (rounds 1-5).
# The process-wide default; RasterContext.xcd_block overrides it for the calls made under a context.
XCD_BLOCK = int(os.environ.get("VIDU4D_SURFEL_XCD_BLOCK", "0"))
# Paired workgroups for the longest tiles of an unsplit forward (VIDU4D_SCHED_PAIR, csrc/blend.hip fwd_pair_walk): K > 0 = the
# tiles longer than K / 4 x the mean list length of the launch are blended by two workgroups each (15: every tile), 0 = off.
# The process-wide default; RasterContext.pair_k overrides it.  Not together with the XCD-local schedule.
# Default 6 (1.5 x the mean; tools/gpu_run.sh pair_ab, profiles/r06_pair_ab.txt): dense Stage-3 ball blend_fwd 348-390 -> 202 us
# (colour + alpha instance) and 389-406 -> 223 us (planes 0-4), bench.py's fitting scene 231 -> 163 and 289 -> 212 us; a launch
# of like lists (the headline scene) pairs nothing.  The rule goes by LIST length: where the long lists of a frame saturate
# early and the long walks are elsewhere the pairs cost their ~1.3 x without shortening the launch (+2-5 % of the forward on
# such a scene, tools/experiments/pair_cost.py).
PAIR_K = int(os.environ.get("VIDU4D_SURFEL_PAIR", "6"))
_cu_count: dict = {}


class RasterContext:
    """The host-side state of ONE caller's rasterizer calls (round 5: it used to be module globals).

    The reference keeps every bit of state in the three buffers a forward returns and is re-entrant
    (rasterize_points.cu:31-37, :92-103).  This build keeps a little more on the host, because it does not stall the GPU in
    the middle of a forward: guesses from the caller's previous frames (pair count, deepest blended list position, longest
    list, long tiles), the forwards whose pair count has not been checked yet (`pending`, deferred mode), one-shot gradient
    output buffers and diagnostics.  Two models stepping alternately, or two threads rendering at once, must not share
    that -- so it lives in a context object the CALLER owns:

        ctx = _C.RasterContext()            # DeformableSurfels owns one (`model.raster_context`)
        with ctx:                           # calls inside use it (a per-thread stack; nests)
            color, radii, allmap = rasterizer(...)

    The autograd functions remember the context of their forward and hand it to their backward (which the autograd engine
    runs on a thread of its own).  The module-level API -- deferred_capacity_check(), check_deferred(), gradient_buffers(),
    hint_scope(), debug_flags(), count_next_walk() -- acts on `current()`: the innermost `with ctx:` of the calling thread, or
    that thread's own default context (the main thread's is the module-level `_capacity_hint`, `_pending`, ... of old)."""

    def __init__(self, scope=None):
        self.scope = scope            # part of the hint keys (kept from rounds 3-4: hint_scope())
        self.deferred = False         # deferred capacity check: forwards do not wait for the pair count ...
        self.graph_mode = False       # ... and record no events (hipGraph capture)
        self.graph_headers = None     # graph_capture_mode: the captured forwards' (device header view, capacity)
        self.pending: list = []       # (event, pinned slot, depth stat, capacity, key, stream) of the unchecked forwards
        self.grad_out: dict = {}      # one-shot caller-supplied gradient outputs (gradient_buffers)
        self.grad_written = None
        self.debug_flags = None       # None: the process-wide DEBUG_FLAGS
        self.xcd_block = None         # None: the process-wide XCD_BLOCK
        self.pair_k = None            # None: the process-wide PAIR_K
        self.walk_counters = None     # (device int64 tensor, one-shot): the next backward counts its tile walk
        self.capacity_hint: dict = {}
        self.depth_hint: dict = {}
        self.len_hint: dict = {}      # longest tile list of earlier frames of a shape (Header word 2; slowly decaying maximum)
        self.long_tiles_hint: dict = {}   # tiles longer than SPLIT_MIN entries (Header word 14), latest frame of a shape
        # Deferred calls also limit the split to the segments the previous frames needed (+25 %, +1): the transmittance
        # pass then skips the tail of long lists that saturate early.  A frame that needed more sets Header::truncated,
        # check_deferred() reports it like an overflow, and the next calls run unlimited.
        self.unlimited: dict = {}
        self.depth_stat: dict = {}    # key -> (device counter, pinned copy)
        self.pinned: dict = {}

    def __enter__(self):
        _stack().append(self)
        return self

    def __exit__(self, *exc):
        _stack().pop()
        return False

    def flags(self) -> int:
        block = XCD_BLOCK if self.xcd_block is None else self.xcd_block
        pair = 0 if block else (PAIR_K if self.pair_k is None else self.pair_k)
        return int(DEBUG_FLAGS if self.debug_flags is None else self.debug_flags) | _lib.sched_xcd_block(block) | _lib.sched_pair(pair)


_tls = threading.local()
_default = RasterContext()   # the main thread's default context


def _stack() -> list:
    st = getattr(_tls, "stack", None)
    if st is None:
        st = _tls.stack = []
    return st


def current() -> RasterContext:
    """The context the calling thread's rasterizer calls use right now."""
    st = _stack()
    if st:
        return st[-1]
    if threading.current_thread() is threading.main_thread():
        return _default
    d = getattr(_tls, "default", None)
    if d is None:
        d = _tls.default = RasterContext()
    return d


# the main thread's default context under its old module-level names (tests and tools read / plant hints through them)
_capacity_hint, _depth_hint, _len_hint = _default.capacity_hint, _default.depth_hint, _default.len_hint
_long_tiles_hint, _unlimited, _pending = _default.long_tiles_hint, _default.unlimited, _default.pending
_depth_stat, _pinned = _default.depth_stat, _default.pinned


@contextlib.contextmanager
def hint_scope(scope):
    """The capacity / split-depth / longest-list hints of the rasterizer calls inside are kept per `scope` (any hashable)
    instead of per image shape alone.  (Rounds 3-4; a caller with a RasterContext of its own needs no scope.)"""
    ctx = current()
    old, ctx.scope = ctx.scope, scope
    try:
        yield
    finally:
        ctx.scope = old


@contextlib.contextmanager
def debug_flags(flags: int, context=None):
    """with debug_flags(_lib.DEBUG_NO_CULL): ... -- forward AND backward inside run with these switches.
    context: the RasterContext they are set on (default: the calling thread's current one).  A model renders under ITS OWN
    context (DeformableSurfels.raster_context): pass `context=model.raster_context` to reach its calls."""
    ctx = context or current()
    old, ctx.debug_flags = ctx.debug_flags, int(flags)
    try:
        yield
    finally:
        ctx.debug_flags = old


def count_next_walk(counters, context=None):
    """Diagnostic: the next rasterize_gaussians_backward call made under `context` (default: the calling thread's current
    one; a model's calls run under `model.raster_context`) adds its tile-walk statistics to `counters` (int64, >=
    _lib.BLEND_STATS entries, on the device, zeroed by the caller)."""
    (context or current()).walk_counters = counters


def _ptr(t):
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


def _check_cuda(*ts):
    for i, t in enumerate(ts):
        if t is not None and t.numel() and not t.is_cuda:
            raise RuntimeError("diff_surfel_rasterization: all tensors must be CUDA/HIP tensors "
                               f"(there is no CPU path in the MI355X build); argument {i} of shape {tuple(t.shape)} "
                               f"is on {t.device}")


class deferred_capacity_check:
    """Context manager: rasterize_gaussians calls inside (under the calling thread's current RasterContext) do not block on
    the pair count.  Opt-in, for callers that can replay a step -- Stage3Trainer: the forward leaves (event, pinned slot,
    capacity) in the context's `pending`, and `check_deferred()` -- called once per step after everything is queued --
    reports whether some frame overflowed its binning buffer (it then rendered only the background) so that the caller can
    discard the step and run it again.  The host can then run a whole step ahead of the GPU."""

    def __enter__(self):
        self._ctx = current()
        self._old, self._ctx.deferred = self._ctx.deferred, True
        return self

    def __exit__(self, *exc):
        self._ctx.deferred = self._old
        return False


class graph_capture_mode:
    """Context manager for hipGraph capture of a step: like deferred_capacity_check, without events; the
    pair-count read-backs go to the per-(stream, index) slots that an eager deferred run on the capture
    stream created before.  `.frames` holds (slot, stat, capacity, key) of the captured forwards; after a
    replay and a stream synchronise pass them to `check_slots`."""

    def __enter__(self):
        self._ctx = c = current()
        self._old = (c.deferred, c.graph_mode)
        c.deferred = c.graph_mode = True
        self._n0 = len(c.pending)
        self.frames = []
        self.headers = c.graph_headers = []   # (header words as a device int32 view, capacity) of the captured forwards
        return self

    def __exit__(self, *exc):
        c = self._ctx
        c.deferred, c.graph_mode = self._old
        self.frames = [(p[1], p[2], p[3], p[4]) for p in c.pending[self._n0:]]
        del c.pending[self._n0:]
        c.graph_headers = None
        return False


HEADER_DEPTH_WORD = 12   # surfel_state.h Header::depth_used
HEADER_LONG_TILES_WORD = 14   # Header::num_long_tiles: tiles longer than SPLIT_MIN (1024) entries -- the same count whether the
#                               frame's forward walked whole tiles or split them (word 4, num_split_pos, follows the mode:
#                               the split decision read it and was bistable -- ADVICE r4)


def _depth_of(stat, slot) -> int:
    """deepest list position the frame blended: from the header copy (this frame's) or the counter pair (the one before)"""
    return int(slot[HEADER_DEPTH_WORD]) if stat == "header" else int(stat[1][0])


def _note_longest_list(ctx, slot, key):
    ctx.len_hint[key] = max(int(slot[2]), int(0.9 * ctx.len_hint.get(key, 0)))
    ctx.long_tiles_hint[key] = int(slot[HEADER_LONG_TILES_WORD])


def auto_split(depth: int, long_tiles: int, compute_units: int) -> bool:
    """VIDU4D_SURFEL_SPLIT=auto: blend this frame's long tiles segment-parallel?  depth: deepest list position a pixel of
    the earlier frames blended; long_tiles: their tiles longer than SPLIT_MIN = 1024 entries (0: not known yet)."""
    return depth > SPLIT_AUTO_LEN and long_tiles < SPLIT_AUTO_TILES_PER_CU * compute_units


def _compute_units(dev) -> int:
    n = _cu_count.get(str(dev))
    if n is None:
        n = _cu_count[str(dev)] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return n


def check_slots(frames, context=None) -> bool:
    ctx = context or current()
    ok = True
    for slot, stat, cap, key in frames:
        n = int(slot[0])
        ctx.capacity_hint[key] = max(ctx.capacity_hint.get(key, 0), int(n * 1.25) + 4096)
        if stat is not None:
            ctx.depth_hint[key] = max(_depth_of(stat, slot), int(0.9 * ctx.depth_hint.get(key, 0)))  # slowly decaying maximum
        _note_longest_list(ctx, slot, key)
        if int(slot[6]):
            ctx.unlimited[key] = 4
            ok = False
        ok = ok and n <= cap
    return ok


def check_deferred(context=None) -> bool:
    """Waits for the pair counts of the deferred forwards queued since the last call (their events were
    recorded right after the tile scan, long passed by the time a step is fully queued), refreshes the
    capacity / split hints, and returns False if any of them overflowed its binning buffer."""
    ctx = context or current()
    ok = True
    for ev, slot, stat, cap, key, _sid in ctx.pending:
        if ev is not None:
            ev.synchronize()
        n = int(slot[0])
        # (a slowly decaying maximum: the buffers shrink again after a prune)
        ctx.capacity_hint[key] = max(int(n * 1.25) + 4096, int(0.98 * ctx.capacity_hint.get(key, 0)))
        if stat is not None:
            ctx.depth_hint[key] = max(_depth_of(stat, slot), int(0.9 * ctx.depth_hint.get(key, 0)))  # slowly decaying maximum
        _note_longest_list(ctx, slot, key)
        if int(slot[6]):  # the segment limit cut a tile short: this frame is incomplete,
            ctx.unlimited[key] = 4   # the next ones run unlimited
            ok = False
        ok = ok and n <= cap
    ctx.pending.clear()
    return ok


def _pinned_slot(device, ctx):
    # one slot per (device, stream): frames rendered concurrently on different streams must not share it;
    # deferred forwards keep theirs until they have been checked, so the k-th pending forward of a stream
    # gets the k-th slot of that stream (the same ones every step)
    sid = torch.cuda.current_stream(device).cuda_stream
    k = sum(1 for p in ctx.pending if p[4][2] == str(device) and p[5] == sid) if ctx.deferred else 0
    key = (str(device), sid, k)
    if key not in ctx.pinned:
        ctx.pinned[key] = torch.zeros(16, dtype=torch.int32).pin_memory()  # Header words 0..15 (surfel_state.h)
    return ctx.pinned[key]


def _set_frame_cams(args, frame_cams, keep):
    """Stacked frames: per-frame (viewmatrix, campos, tan_fovx, tan_fovy) into the argument struct."""
    args.frames = len(frame_cams)
    for f, (vm, cp, tfx, tfy) in enumerate(frame_cams):
        vm, cp = _f32c(vm, "viewmatrix"), _f32c(cp, "campos")
        keep += [vm, cp]
        args.frame_viewmatrix[f], args.frame_campos[f] = vm.data_ptr(), cp.data_ptr()
        args.frame_tan_fovx[f], args.frame_tan_fovy[f] = float(tfx), float(tfy)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, frame_cams=None, sh_rest=None, raw_params=False, aux_planes=0, context=None):
    """-> (num_rendered, out_color, out_others, radii, geomBuffer, binningBuffer, imgBuffer)

    context: the RasterContext whose hints / deferred mode / debug switches this call uses (default: current()).

    sh_rest / raw_params (extension: the canonical parameters as the optimizer holds them, no activation / concatenation
    launches in between): with sh_rest (P,15,3), `sh` is the (P,1,3) DC tensor; raw_params: `scales` are log-scales and
    `opacity` logits (exp / sigmoid applied by the kernels, gs/scene/gaussian_model.py:47-57, :98-118).
    aux_planes (extension): bit mask of the out_others planes the caller reads, 0 = all; _lib.AUX_ALPHA (plane 1 only)
    selects the colour + alpha blend (the other planes come out as zeros; hand the same mask to the backward).

    frame_cams (extension): a list of F <= 8 (viewmatrix, campos, tan_fovx, tan_fovy) -- F frames that
    share opacity / scales / sh are rasterized by one launch set: means3D (F,P,3), rotations (F,P,4) -> out_color
    (3,F,H,W), out_others (8,F,H,W), radii (F,P); viewmatrix / campos / tan_fov* arguments are ignored."""
    lib = _lib.load()
    ctx = context or current()
    _deferred, _graph_mode = ctx.deferred, ctx.graph_mode
    F = 1 if frame_cams is None else len(frame_cams)
    if frame_cams is not None:
        if not 1 <= F <= 8:
            raise RuntimeError("frame_cams: 1 to 8 frames")
        if means3D.ndim != 3 or means3D.shape[0] != F or rotations.ndim != 3 or rotations.shape[0] != F:
            raise RuntimeError("stacked frames: means3D (F, num_points, 3) and rotations (F, num_points, 4) required")
        means3D, rotations = means3D.reshape(-1, 3), rotations.reshape(-1, 4)
        viewmatrix, campos, tan_fovx, tan_fovy = frame_cams[0]
        if projmatrix is None or not projmatrix.numel():
            projmatrix = viewmatrix
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if scales.ndim != 2 or scales.shape[1] != 2:
        raise RuntimeError("scales must have dimensions (num_points, 2)")
    if rotations.ndim != 2 or rotations.shape[1] != 4:
        raise RuntimeError("rotations must have dimensions (num_points, 4)")
    _check_cuda(background, means3D, colors, opacity, scales, rotations, transMat_precomp, viewmatrix, projmatrix, sh,
                campos)
    dev = means3D.device
    P, H, W = means3D.shape[0] // F, int(image_height), int(image_width)  # (P: surfels per frame)
    means3D = _f32c(means3D, "means3D")
    scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations")
    opacity = _f32c(opacity, "opacity")
    background = _f32c(background, "background")
    viewmatrix = _f32c(viewmatrix, "viewmatrix")
    projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos")
    M = 0
    if sh.numel():
        sh = _f32c(sh, "sh")
        M = sh.shape[1]
    if sh_rest is not None:
        sh_rest = _f32c(sh_rest, "sh_rest")
        if sh.shape[1:] != (1, 3) or sh_rest.shape[1:] != (15, 3) or sh_rest.shape[0] != sh.shape[0]:
            raise RuntimeError("canonical SH pair: sh (num_points, 1, 3) and sh_rest (num_points, 15, 3) required")
        M = 16
    if colors.numel():
        colors = _f32c(colors, "colors")

    plane = (H, W) if frame_cams is None else (F, H, W)
    out_color = torch.empty((3,) + plane, dtype=torch.float32, device=dev)
    out_others = torch.empty((8,) + plane, dtype=torch.float32, device=dev)
    radii = torch.empty((P,) if frame_cams is None else (F, P), dtype=torch.int32, device=dev)
    geom = torch.empty((lib.vidu4d_surfel_geom_bytes(P * F),), dtype=torch.uint8, device=dev)
    img = torch.empty((lib.vidu4d_surfel_image_bytes_frames(W, H, F),), dtype=torch.uint8, device=dev)

    a = _lib.ForwardArgs()
    keep = []
    if frame_cams is not None and F > 1:
        _set_frame_cams(a, frame_cams, keep)
    a.P, a.D, a.M, a.width, a.height = P, int(degree), M, W, H
    a.tan_fovx, a.tan_fovy = float(tan_fovx), float(tan_fovy)
    a.scale_modifier = float(scale_modifier)
    a.prefiltered, a.debug = int(bool(prefiltered)), int(bool(debug))
    a.background, a.means3D, a.shs, a.colors_precomp = _ptr(background), _ptr(means3D), _ptr(sh), _ptr(colors)
    if sh_rest is not None:
        a.shs, a.sh_dc, a.sh_rest = None, _ptr(sh), _ptr(sh_rest)
    a.raw_params, a.aux_planes, a.debug_flags = int(bool(raw_params)), int(aux_planes), ctx.flags()
    a.opacities, a.scales, a.rotations = _ptr(opacity), _ptr(scales), _ptr(rotations)
    a.transMat_precomp = _ptr(transMat_precomp)
    a.viewmatrix, a.projmatrix, a.campos = _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos)
    a.out_color, a.out_others, a.radii = out_color.data_ptr(), out_others.data_ptr(), _ptr(radii)
    a.geom_buffer, a.geom_bytes = geom.data_ptr(), geom.numel()
    a.image_buffer, a.image_bytes = img.data_ptr(), img.numel()
    stream = _stream(dev)
    hint = None
    # Hints are keyed on the image shape and device only: the surfel count changes with every densify / prune,
    # and the pair count, split depth and per-stream counters of the previous frames stay good guesses across it
    # (keying on P leaked one entry -- with a device tensor and a pinned buffer -- per surfel count).
    # ... and, where the caller names one, on its scope (hint_scope: a model instance) -- two models that render the same image
    # size no longer feed each other's guesses.  (key[2] stays the device: _pinned_slot reads it.)
    key = (W, H, str(dev), F, ctx.scope)
    stat = None
    hint = ctx.capacity_hint.get(key)
    if _SPLIT == "auto":
        depth = ctx.depth_hint.get(key, 0)
        a.segment_split = int(auto_split(depth, ctx.long_tiles_hint.get(key, 0), _compute_units(dev)))
        if a.segment_split and _deferred and not debug:
            if ctx.unlimited.get(key, 0) > 0:
                ctx.unlimited[key] -= 1
            else:
                a.segment_split = max(2, (int(depth * 1.25) + SEG_LEN - 1) // SEG_LEN + 1)
        if _deferred and not debug and hint is not None and not _EXACT:
            # (the header is read back after the blend anyway: the depth counter lives in it -- word 12, zeroed by the
            # projection kernel -- instead of in a tensor of its own that wants a reset and a copy per frame)
            stat = "header"
            a.depth_used = geom.data_ptr() + 4 * HEADER_DEPTH_WORD
        else:
            stat = ctx.depth_stat.get((key, stream))
            if stat is None:
                stat = ctx.depth_stat[(key, stream)] = (torch.zeros(1, dtype=torch.int32, device=dev),
                                                     torch.zeros(1, dtype=torch.int32).pin_memory())
            a.depth_used = stat[0].data_ptr()
    else:
        a.segment_split = int(_SPLIT == "1")
    # long lists: the MSD split + bucket sorts from ~10 k entries per list on, one workgroup per list through global memory
    # below (dense Stage-3 ball, 5 k-entry lists: 81 us against 109; 3 %-coverage object, 25 k: 171 against 78)
    a.long_list_sort = 0 if ctx.len_hint.get(key, 1 << 30) >= MSD_SORT_FROM else 1
    a.max_list_hint = int(ctx.len_hint.get(key, 0))   # (0: unknown.  Short lists only: one sort launch instead of two)
    if a.segment_split and ((int(aux_planes) == _lib.AUX_ALPHA and _SPEC) or
                            (int(aux_planes) != _lib.AUX_ALPHA and int(aux_planes) and not (int(aux_planes) & ~_lib.AUX_GEOM)
                             and _SPEC_GEOM)):
        a.assume_unsaturated = 1   # (planes 0-4 at most: no median sample, nothing depends on the start transmittance)

    if P == 0:  # rasterize_points.cu:105: nothing is launched, outputs are zeros
        out_color.zero_()
        out_others.zero_()
        binning = torch.empty((0,), dtype=torch.uint8, device=dev)
        return 0, out_color, out_others, radii, geom, binning, img

    _lib.check(lib.vidu4d_surfel_forward_plan(C.byref(a), stream), "surfel forward (plan)")
    R = C.c_int64(0)
    if _EXACT or hint is None or debug:
        _lib.check(lib.vidu4d_surfel_num_rendered(C.byref(a), stream, C.byref(R)), "surfel forward (count)")
        cap = max(int(R.value), 1)
        binning = torch.empty((lib.vidu4d_surfel_binning_bytes(cap),), dtype=torch.uint8, device=dev)
        _lib.check(lib.vidu4d_surfel_forward_run(C.byref(a), binning.data_ptr(), binning.numel(), cap, stream),
                   "surfel forward (run)")
        num_rendered = int(R.value)
    else:
        cap = hint
        binning = torch.empty((lib.vidu4d_surfel_binning_bytes(cap),), dtype=torch.uint8, device=dev)
        slot = _pinned_slot(dev, ctx)
        if stat is not None and stat != "header":  # depth reached by the previous frame on this stream; then reset for this one
            stat[1].copy_(stat[0], non_blocking=True)
            stat[0].zero_()
        if _deferred and not debug:
            # nobody waits for the header before the step is fully queued: read it back after the blend, so
            # that it also carries the `truncated` flag of a segment-limited split
            _lib.check(lib.vidu4d_surfel_forward_run(C.byref(a), binning.data_ptr(), binning.numel(), cap, stream),
                       "surfel forward (run)")
            slot.copy_(geom[:64].view(torch.int32), non_blocking=True)
            ev = None
            if not _graph_mode:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
            elif ctx.graph_headers is not None:
                # (a captured step cannot ask the host for its verdict: lab4d/captured_step.py derives a device word from
                # the header -- pair count against this capacity, the truncated flag -- that the step's updates read)
                ctx.graph_headers.append((geom[:64].view(torch.int32), cap))
            ctx.pending.append((ev, slot, stat, cap, key, stream))
            binning._vidu4d_capacity = cap
            binning._vidu4d_split = int(a.segment_split)
            return cap, out_color, out_others, radii, geom, binning, img
        slot.copy_(geom[:64].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        _lib.check(lib.vidu4d_surfel_forward_run(C.byref(a), binning.data_ptr(), binning.numel(), cap, stream),
                   "surfel forward (run)")
        ev.synchronize()  # waits for preprocess + scan only; sort and blend keep running
        num_rendered = int(slot[0])
        if stat is not None:
            ctx.depth_hint[key] = max(int(stat[1][0]), int(0.9 * ctx.depth_hint.get(key, 0)))  # slowly decaying maximum
        if num_rendered > cap:  # guess too small: queue the tail again with an exact buffer
            cap = num_rendered
            binning = torch.empty((lib.vidu4d_surfel_binning_bytes(cap),), dtype=torch.uint8, device=dev)
            _lib.check(lib.vidu4d_surfel_forward_run(C.byref(a), binning.data_ptr(), binning.numel(), cap, stream),
                       "surfel forward (re-run)")
    ctx.capacity_hint[key] = max(int(num_rendered * 1.25) + 4096, 4096)
    # the capacity the buffers were carved with travels to backward inside the buffer tensor
    binning._vidu4d_capacity = cap
    binning._vidu4d_split = int(a.segment_split)
    return num_rendered, out_color, out_others, radii, geom, binning, img


# ---- caller-supplied gradient outputs.  The backward writes every gradient of the SHARED parameters (opacity, scales, SH)
# exactly once, so a caller that keeps a persistent gradient buffer -- the flat exchange buffer of the frame-parallel
# path (lab4d/stage3.py, bench.py) -- can have them written there instead of into fresh tensors: autograd then adopts the
# returned tensor as `.grad` (no accumulation pass over a pre-bound, zero-filled .grad: 2 x 46 MB per step at 200 k
# surfels).  One-shot: the first backward call inside the context takes the buffers.  Process-wide state of a
# single-threaded caller, like the capacity hints above.
@contextlib.contextmanager
def gradient_buffers(on_written=None, context=None, **buffers):
    """buffers: dL_dopacity (P,1), dL_dscales (P,2), dL_dsh (P,M,3) or dL_dsh_dc (P,1,3) / dL_dsh_rest (P,15,3): contiguous
    fp32 tensors on the device, 16-byte aligned.  on_written: called (once) right after the backward that took the buffers
    has been queued -- the place to record the event a collective over them waits for, while the rest of the step's
    backward is still to come.  State of `context` (default: the calling thread's current RasterContext): the backward of a
    forward made under that context takes them."""
    ctx = context or current()
    old = (ctx.grad_out, ctx.grad_written)
    ctx.grad_out, ctx.grad_written = {k: v for k, v in buffers.items() if v is not None}, on_written
    try:
        yield
    finally:
        ctx.grad_out, ctx.grad_written = old


def _grad_output(ctx, name, shape, opt):
    t = ctx.grad_out.pop(name, None)
    if (t is not None and t.numel() == math.prod(shape) and t.is_contiguous() and t.dtype == torch.float32
            and t.device == opt["device"] and t.data_ptr() % 16 == 0):
        return t.view(shape)   # (a fresh tensor object over the caller's memory: autograd may adopt it as .grad)
    return torch.empty(shape, **opt)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_others, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                 binning_capacity=None, segment_split=None, frame_cams=None, sh_rest=None,
                                 raw_params=False, aux_planes=0, context=None):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations)

    sh_rest / raw_params as in rasterize_gaussians: dL_dsh is then the pair (dL_dsh_dc (P,1,3), dL_dsh_rest (P,15,3)),
    dL_dscales / dL_dopacity are w.r.t. the log-scales / logits.

    frame_cams: as in rasterize_gaussians (means3D (F,P,3), rotations (F,P,4), radii (F,P), dL_dout_color (3,F,H,W),
    dL_dout_others (8,F,H,W)); per-frame gradients come back (F,P,.), those of opacity / scales / sh summed over the
    frames."""
    lib = _lib.load()
    ctx = context or current()   # (the autograd functions pass their forward's: the engine's thread has no `with ctx:`)
    F = 1 if frame_cams is None else len(frame_cams)
    if frame_cams is not None:
        means3D, rotations = means3D.reshape(-1, 3), rotations.reshape(-1, 4)
        viewmatrix, campos, tan_fovx, tan_fovy = frame_cams[0]
        if projmatrix is None or not projmatrix.numel():
            projmatrix = viewmatrix
    _check_cuda(background, means3D, radii, colors, scales, rotations, viewmatrix, projmatrix, sh, campos, geomBuffer,
                binningBuffer, imageBuffer)
    dev = means3D.device
    P = means3D.shape[0] // F  # (surfels per frame)
    H, W = dL_dout_color.shape[-2], dL_dout_color.shape[-1]
    M = sh.shape[1] if sh.numel() else 0
    if sh_rest is not None:
        M = 16
    opt = dict(dtype=torch.float32, device=dev)
    lead = (P,) if frame_cams is None else (F, P)
    dL_dmeans3D = torch.empty(lead + (3,), **opt)
    dL_dmeans2D = torch.empty(lead + (3,), **opt)
    dL_dcolors = torch.empty(lead + (3,), **opt)
    n_buffers = len(ctx.grad_out)
    dL_dopacity = _grad_output(ctx, "dL_dopacity", (P, 1), opt)
    dL_dtransMat = torch.empty(lead + (9,), **opt)
    if sh_rest is not None:
        dL_dsh = (_grad_output(ctx, "dL_dsh_dc", (P, 1, 3), opt), _grad_output(ctx, "dL_dsh_rest", (P, 15, 3), opt))
    else:
        dL_dsh = _grad_output(ctx, "dL_dsh", (P, M, 3), opt)
    dL_dscales = _grad_output(ctx, "dL_dscales", (P, 2), opt)
    took_buffers = len(ctx.grad_out) < n_buffers
    dL_drotations = torch.empty(lead + (4,), **opt)
    if P == 0:
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations
    if binning_capacity is None:
        binning_capacity = getattr(binningBuffer, "_vidu4d_capacity", None)
    if binning_capacity is None:
        binning_capacity = max(int(R), 1)
    ws = torch.empty((lib.vidu4d_surfel_backward_workspace_bytes(P * F),), dtype=torch.uint8, device=dev)
    means3D = _f32c(means3D, "means3D")
    scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations")
    dL_dout_color = _f32c(dL_dout_color, "dL_dout_color")
    dL_dout_others = _f32c(dL_dout_others, "dL_dout_others")
    background = _f32c(background, "background")
    viewmatrix = _f32c(viewmatrix, "viewmatrix")
    projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos")
    if sh.numel():
        sh = _f32c(sh, "sh")
    if sh_rest is not None:
        sh_rest = _f32c(sh_rest, "sh_rest")
    if colors.numel():
        colors = _f32c(colors, "colors")

    b = _lib.BackwardArgs()
    keep = []
    if frame_cams is not None and F > 1:
        _set_frame_cams(b, frame_cams, keep)
    b.P, b.D, b.M, b.width, b.height = P, int(degree), M, W, H
    b.tan_fovx, b.tan_fovy, b.scale_modifier, b.debug = float(tan_fovx), float(tan_fovy), float(scale_modifier), int(
        bool(debug))
    b.background, b.means3D, b.radii = _ptr(background), _ptr(means3D), _ptr(radii)
    b.shs, b.colors_precomp, b.scales, b.rotations = _ptr(sh), _ptr(colors), _ptr(scales), _ptr(rotations)
    b.transMat_precomp = _ptr(transMat_precomp)
    b.viewmatrix, b.projmatrix, b.campos = _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos)
    b.dL_dout_color, b.dL_dout_others = dL_dout_color.data_ptr(), dL_dout_others.data_ptr()
    b.geom_buffer, b.binning_buffer, b.image_buffer = geomBuffer.data_ptr(), _ptr(binningBuffer), imageBuffer.data_ptr()
    b.binning_capacity = int(binning_capacity)
    if segment_split is None:
        segment_split = getattr(binningBuffer, "_vidu4d_split", 0)
    b.segment_split = int(segment_split)
    b.workspace, b.workspace_bytes = ws.data_ptr(), ws.numel()
    b.dL_dmeans2D, b.dL_dcolors, b.dL_dopacity = dL_dmeans2D.data_ptr(), dL_dcolors.data_ptr(), dL_dopacity.data_ptr()
    b.dL_dmeans3D, b.dL_dtransMat = dL_dmeans3D.data_ptr(), dL_dtransMat.data_ptr()
    if sh_rest is not None:
        b.shs, b.sh_dc, b.sh_rest = None, _ptr(sh), _ptr(sh_rest)
        b.dL_dsh, b.dL_dsh_dc, b.dL_dsh_rest = None, dL_dsh[0].data_ptr(), dL_dsh[1].data_ptr()
    else:
        b.dL_dsh = _ptr(dL_dsh)
    b.raw_params, b.aux_planes, b.debug_flags = int(bool(raw_params)), int(aux_planes), ctx.flags()
    if ctx.walk_counters is not None:
        b.diag_walk_counters, keep_counters, ctx.walk_counters = ctx.walk_counters.data_ptr(), ctx.walk_counters, None  # noqa: F841
    b.dL_dscales, b.dL_drotations = dL_dscales.data_ptr(), dL_drotations.data_ptr()
    _lib.check(lib.vidu4d_surfel_backward(C.byref(b), _stream(dev)), "surfel backward")
    if ctx.grad_written is not None and took_buffers:
        cb, ctx.grad_written = ctx.grad_written, None
        cb()
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool (P,) (rasterize_points.cu:242-261)"""
    lib = _lib.load()
    _check_cuda(means3D, viewmatrix, projmatrix)
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P:
        m = _f32c(means3D, "means3D")
        v = _f32c(viewmatrix, "viewmatrix")
        pj = _f32c(projmatrix, "projmatrix")
        _lib.check(lib.vidu4d_surfel_mark_visible(P, m.data_ptr(), v.data_ptr(), pj.data_ptr(), present.data_ptr(),
                                                  _stream(means3D.device)), "mark_visible")
    return present


def read_state(what: str, fwd_inputs: dict, geomBuffer, binningBuffer, imgBuffer, P, W, H, dtype, max_count, frames=1):
    """Test helper: copies one internal array (see _lib.STATE) to a host tensor (frames > 1: P per frame)."""
    lib = _lib.load()
    a = _lib.ForwardArgs()
    a.P, a.width, a.height, a.frames = P, W, H, frames
    a.geom_buffer, a.geom_bytes = geomBuffer.data_ptr(), geomBuffer.numel()
    a.image_buffer, a.image_bytes = imgBuffer.data_ptr(), imgBuffer.numel()
    cap = getattr(binningBuffer, "_vidu4d_capacity", 0)
    dst = torch.empty((max(max_count, 1),), dtype=dtype)
    n = C.c_int64(0)
    _lib.check(lib.vidu4d_surfel_state_read(C.byref(a), _ptr(binningBuffer), cap, _lib.STATE[what], dst.data_ptr(),
                                            dst.numel() * dst.element_size(), C.byref(n),
                                            _stream(geomBuffer.device)), f"state_read({what})")
    return dst[: n.value].clone()
