"""Decision replay (TEST INFRASTRUCTURE, like everything under oracle/): explains the differences between the CPU oracle and
another implementation of the blend -- the reference's own sources compiled for gfx950 (oracle/_ref), or the HIP product --
as THRESHOLD FLIPS instead of budgeting them.

Two fp32 implementations of forward.cu:265-463 / backward.cu:143-449 differ by rounding (~1e-6..1e-5 of a tensor's scale) and
by per-(pixel, list entry) decisions on values within rounding of a threshold: the accept test (alpha >= 1/255, depth >= 0.2:
forward.cu:385-395), the rho3d <= rho2d branch (:379-383), the end of the walk T (1 - alpha) < 1e-4 (:400-405), the median
sample T > 0.5 (:416-421).  `explain()`
  1. finds the pixels on which the two disagree (any forward plane beyond `tol` of its scale, or another n_contrib);
  2. takes the other implementation's walk end and median sample for those pixels from ITS n_contrib (both are outputs);
  3. searches, per pixel, the smallest set of near-threshold pairs (oracle_pixel_candidates: within `cand_tol` of a
     threshold, relative) whose accept / branch decision, inverted, makes the oracle's pixel equal the other's;
  4. recomputes the WHOLE forward and backward with those decisions forced (oracle_render_{forward,backward}_replay + the
     per-surfel chain) -- so that a test can hold EVERY entry of EVERY tensor to `tol` of scale, with no outlier budget.
"""
from __future__ import annotations

import ctypes as C
import itertools

import numpy as np

from . import surfel_oracle as so

FREE = 0xFFFFFFFF
DIST_ATOL = 2e-6   # plane 6 (distortion): a difference of O(1) fp32 sums, compared on its absolute noise floor (tests/util.py)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _pixel_outputs(st, feats, bg, pixel, f_last, f_med, flips, strict=0):
    L = so.lib()
    pos = _u32([f[0] for f in flips]) if flips else np.zeros(1, np.uint32)
    what = _u32([f[1] for f in flips]) if flips else np.zeros(1, np.uint32)
    out = np.zeros(14, np.float32)
    n2 = np.zeros(2, np.uint32)
    L.oracle_replay_pixel(C.c_int(st["W"]), C.c_int(st["H"]), so._p(st["ranges"]), so._p(st["point_list"]), so._p(st["means2D"]),
                          so._p(feats), so._p(st["transMat"]), so._p(st["normal_opacity"]), so._p(bg), C.c_uint32(int(pixel)),
                          C.c_uint32(int(f_last)), C.c_uint32(int(f_med)), C.c_uint32(len(flips)), so._p(pos), so._p(what),
                          C.c_int(int(strict)), so._p(out), so._p(n2))
    return out, n2


def _candidates(st, pixel, max_pos, cand_tol, strict=0, max_out=64):
    L = so.lib()
    L.oracle_pixel_candidates.restype = C.c_uint32
    pos, kind, margin = np.zeros(max_out, np.uint32), np.zeros(max_out, np.uint32), np.zeros(max_out, np.float32)
    n = L.oracle_pixel_candidates(C.c_int(st["W"]), C.c_int(st["H"]), so._p(st["ranges"]), so._p(st["point_list"]),
                                  so._p(st["means2D"]), so._p(st["transMat"]), so._p(st["normal_opacity"]), C.c_uint32(int(pixel)),
                                  C.c_uint32(int(max_pos)), C.c_float(cand_tol), C.c_float(cand_tol), C.c_int(int(strict)),
                                  C.c_uint32(max_out), so._p(pos), so._p(kind), so._p(margin))
    n = min(int(n), max_out)
    order = np.argsort(margin[:n])
    return [(int(pos[i]), int(kind[i]), float(margin[i])) for i in order]


def explain(st, other_color, other_others, other_n_contrib, tol=1e-4, cand_tol=2e-3, max_flips=3, max_candidates=10, strict=0):
    """st: the oracle's forward state (surfel_oracle.forward); other_*: the other implementation's color (3,H,W), others
    (8,H,W), n_contrib (2,H,W).  strict: 1 when the other implementation evaluates the (pixel, surfel) pairs in the source's
    unfused operation order (oracle/_ref's strict build) -- the oracle's side of the comparison is then ITS blend in that
    order (oracle_render_forward_replay with nothing forced), not the explicit-FMA sequence it shares with the product.
    -> dict(pixels, explained, unexplained (list of pixel ids), flips [(pixel, pos, what)], forced_last, forced_median,
    max_margin, by_kind, strict)."""
    W, H = st["W"], st["H"]
    HW = W * H
    inp = st["_inputs"]
    feats = inp["colors_precomp"] if inp["colors_precomp"] is not None else st["rgb"]
    bg = inp["bg"]
    oc = np.asarray(other_color, np.float32).reshape(3, HW)
    oo = np.asarray(other_others, np.float32).reshape(8, HW)
    on = np.asarray(other_n_contrib).astype(np.uint32).reshape(2, HW)
    if strict:
        base, _ = replay(st, dict(flips=[], forced_last=None, forced_median=None, strict=1), None, None)
    else:
        base = st
    mc, mo, mn = base["color"].reshape(3, HW), base["others"].reshape(8, HW), base["n_contrib"].reshape(2, HW)
    s_color = float(np.abs(oc).max()) + 1e-30
    s_others = np.abs(oo).max(axis=1) + 1e-30
    tol_c = tol * s_color
    tol_o = tol * s_others
    tol_o[6] = tol_o[6] + DIST_ATOL
    has = on[0] > 0
    diff = (np.abs(mc - oc) > tol_c).any(axis=0) | (np.abs(mo - oo) > tol_o[:, None]).any(axis=0) | (mn[0] != on[0]) | \
        ((mn[1] != on[1]) & has & (mn[0] > 0))
    pixels = np.nonzero(diff)[0]
    forced_last = np.full(HW, FREE, np.uint32)
    forced_median = np.full(HW, FREE, np.uint32)
    flips, unexplained, max_margin, by_kind = [], [], 0.0, {"walk_end_or_median_only": 0, "accept": 0, "branch": 0}

    def matches(out, p):
        return (np.abs(out[:3] - oc[:, p]) <= tol_c).all() and (np.abs(out[3:11] - oo[:, p]) <= tol_o).all()

    for p in pixels:
        fl = int(on[0, p])
        fm = int(on[1, p]) if fl > 0 else 0
        forced_last[p], forced_median[p] = fl, fm
        out, _ = _pixel_outputs(st, feats, bg, p, fl, fm, [], strict)
        if matches(out, p):
            by_kind["walk_end_or_median_only"] += 1
            continue
        cands = _candidates(st, p, max(fl, int(mn[0, p])) + 2, cand_tol, strict)[:max_candidates]
        found = None
        for k in range(1, max_flips + 1):
            for combo in itertools.combinations(cands, k):
                fs = {}
                for pos, kind, _m in combo:
                    fs[pos] = fs.get(pos, 0) | (1 if kind == 1 else 2)
                out, _ = _pixel_outputs(st, feats, bg, p, fl, fm, sorted(fs.items()), strict)
                if matches(out, p):
                    found = (combo, fs)
                    break
            if found:
                break
        if found is None:
            unexplained.append(int(p))
            forced_last[p], forced_median[p] = FREE, FREE
            continue
        combo, fs = found
        for pos, what in sorted(fs.items()):
            flips.append((int(p), pos, what))
        for _pos, kind, m in combo:
            max_margin = max(max_margin, m)
            by_kind["accept" if kind == 1 else "branch"] += 1
    flips.sort()
    return dict(pixels=int(len(pixels)), explained=int(len(pixels) - len(unexplained)), unexplained=unexplained, flips=flips,
                forced_last=forced_last, forced_median=forced_median, max_margin=max_margin, by_kind=by_kind, strict=int(strict))


def replay(st, ex, dL_dcolor, dL_dothers):
    """The whole forward and backward of the oracle with the decisions of `ex` (explain()) forced, in the operation order
    `ex["strict"]` names.  dL_dcolor None: the forward only.
    -> (state dict with color / others / final_T / n_contrib replaced, gradient dict as surfel_oracle.backward)."""
    L = so.lib()
    W, H, P = st["W"], st["H"], st["P"]
    inp = st["_inputs"]
    feats = inp["colors_precomp"] if inp["colors_precomp"] is not None else st["rgb"]
    fp = _u32([f[0] for f in ex["flips"]]) if ex["flips"] else np.zeros(1, np.uint32)
    fpos = _u32([f[1] for f in ex["flips"]]) if ex["flips"] else np.zeros(1, np.uint32)
    fwhat = _u32([f[2] for f in ex["flips"]]) if ex["flips"] else np.zeros(1, np.uint32)
    nf = len(ex["flips"])
    out = dict(st)
    out["final_T"] = np.zeros((3, H, W), np.float32)
    out["n_contrib"] = np.zeros((2, H, W), np.uint32)
    out["color"] = np.zeros((3, H, W), np.float32)
    out["others"] = np.zeros((8, H, W), np.float32)
    L.oracle_render_forward_replay(C.c_int(W), C.c_int(H), so._p(st["ranges"]), so._p(st["point_list"]), so._p(st["means2D"]),
                                   so._p(feats), so._p(st["transMat"]), so._p(st["normal_opacity"]), so._p(inp["bg"]),
                                   so._p(ex["forced_last"]), so._p(ex["forced_median"]), C.c_uint32(nf), so._p(fp), so._p(fpos),
                                   so._p(fwhat), C.c_int(int(ex.get("strict", 0))), so._p(out["final_T"]), so._p(out["n_contrib"]),
                                   so._p(out["color"]), so._p(out["others"]))
    if dL_dcolor is None:
        return out, None
    dL_dcolor = so._f32(dL_dcolor, (3, H, W))
    dL_dothers = so._f32(dL_dothers, (8, H, W))
    acc_T, acc_m2d = np.zeros((P, 9), np.float64), np.zeros((P, 3), np.float64)
    acc_n, acc_o, acc_c = np.zeros((P, 3), np.float64), np.zeros((P,), np.float64), np.zeros((P, 3), np.float64)
    L.oracle_render_backward_replay(C.c_int(W), C.c_int(H), so._p(st["ranges"]), so._p(st["point_list"]), so._p(inp["bg"]),
                                    so._p(st["means2D"]), so._p(st["normal_opacity"]), so._p(st["transMat"]), so._p(feats),
                                    so._p(out["final_T"]), so._p(out["n_contrib"]), so._p(dL_dcolor), so._p(dL_dothers),
                                    C.c_uint32(nf), so._p(fp), so._p(fpos), so._p(fwhat), C.c_int(int(ex.get("strict", 0))), so._p(acc_T),
                                    so._p(acc_m2d), so._p(acc_n), so._p(acc_o), so._p(acc_c))
    g = dict(dL_dtransMat=acc_T.astype(np.float32), dL_dmeans2D=acc_m2d.astype(np.float32), dL_dnormal=acc_n.astype(np.float32),
             dL_dopacity=acc_o.astype(np.float32).reshape(P, 1), dL_dcolors=acc_c.astype(np.float32))
    g["dL_dmeans2D_filter"] = g["dL_dmeans2D"].copy()
    g["dL_dtransMat_render"] = g["dL_dtransMat"].copy()
    g.update(so.backward_chain(out, g["dL_dtransMat_render"], g["dL_dmeans2D_filter"], g["dL_dnormal"], g["dL_dcolors"]))
    return out, g
