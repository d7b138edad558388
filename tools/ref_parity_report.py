#!/usr/bin/env python
"""MEASURED agreement between the CPU oracle, the product and the reference's own sources on an MI355X
(TEST INFRASTRUCTURE; uses oracle/_ref built by oracle/ref_build/build_ref.py in both variants:
"strict" = fp contraction off + exact rsqrt, "default" = hipcc's own contraction).

For each configuration it prints and stores (gpurun_out/ref_parity.json -> copied to profiles/):
  integers  radii / tiles_touched / point_list / sorted keys / ranges / n_contrib: number of differing entries
  floats    per tensor: fraction of entries whose error exceeds 1e-4 of the tensor's scale, and the worst error;
            round 5: the relative L2 error, and the element-wise relative error (median / p99 / max) over the entries whose
            reference magnitude is above 1e-3 of the tensor's scale
for the pairs  oracle vs strict reference, oracle vs default reference, product vs strict, product vs default.
The budgets of tests/test_gpu_reference.py are set from these numbers (<= 2x measured).

Usage (GPU box):  python tools/ref_parity_report.py [--configs small,cfgA,cfgB,cfgE_slice]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import surfel_oracle as so  # noqa: E402
from oracle.ref_build import ref  # noqa: E402
from tests.util import oracle_forward, relative_error_stats, to_np  # noqa: E402
from vidu4d_amd.synthetic import make_scene, make_upstream_grads  # noqa: E402

CONFIGS = {
    "mid": dict(n=5000, width=128, height=128, seed=11),
    "cfgA": dict(n=50_000, width=256, height=256, seed=1234),
    "cfgB": dict(n=200_000, width=512, height=512, seed=1234),
    # BASELINE.json configs[4] is 1M surfels at 1920x1080; the CPU oracle walks it in ~1 min, so the report
    # uses a 250k-surfel slice of it at the full resolution (partial tiles: 1080 is not a multiple of 16)
    "cfgE_slice": dict(n=250_000, width=1920, height=1080, seed=1234),
    "cfgE_full": dict(n=1_000_000, width=1920, height=1080, seed=1234),   # configs[4] itself (round 4)
}
from tests.util import CASES as SMALL_CASES  # noqa: E402
for _n in ("tiny", "ragged", "small", "deg1", "subpixel", "huge", "init_opacity"):
    CONFIGS[_n] = SMALL_CASES[_n]
FLOAT_GRADS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")


def float_stats(got, want, rtol=1e-4):
    got, want = to_np(got).astype(np.float64), to_np(want).astype(np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want)
    out = {"outlier_frac": float((err > rtol * scale).mean()), "worst_rel": float(err.max() / scale),
           "finite": bool(np.isfinite(got).all())}
    out.update(relative_error_stats(got, want))   # (round 5: relative L2, element-wise relative p50 / p99 / max)
    return out


def int_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return {"differs": "shape", "a": list(a.shape), "b": list(b.shape)}
    return int((a != b).sum())


def n_contrib_diff(ours, ref_n):
    """Differing entries; the median plane only where the pixel has contributors (without any, the reference
    converts its initial float -1 to uint32: undefined behaviour, garbage in the strict build, never read)."""
    has = ref_n[0] > 0
    return int((ref_n[0] != ours[0]).sum() + (ref_n[1] != ours[1])[has].sum())


def compare_images(out, others, grads, ref_fwd, ref_grads):
    res = {"color": float_stats(out, ref_fwd["color"])}
    for i in range(8):
        res[f"others{i}"] = float_stats(others[i], ref_fwd["others"][i])
    for k in FLOAT_GRADS:
        res[k] = float_stats(grads[k], ref_grads[k])
    return res


def run_product(d, dc, do, W, Hh):
    import diff_surfel_rasterization as dsr
    from vidu4d_amd import _C
    rs = dsr.GaussianRasterizationSettings(Hh, W, d.tanfovx, d.tanfovy, d.bg, 1.0, d.viewmatrix, d.projmatrix,
                                           d.sh_degree, d.campos, False, False)
    leaves = [t.clone().requires_grad_(True) for t in (d.means3D, d.opacities, d.scales, d.rotations, d.shs)]
    m2d = torch.zeros_like(leaves[0], requires_grad=True)
    # through the native functions so that the internal state can be read back
    st = _C.rasterize_gaussians(d.bg, leaves[0], torch.empty(0, device=d.bg.device), leaves[1], leaves[2], leaves[3], 1.0,
                                torch.empty(0, device=d.bg.device), d.viewmatrix, d.projmatrix, d.tanfovx, d.tanfovy,
                                Hh, W, leaves[4], d.sh_degree, d.campos, False, False)
    R, color, others, radii, geom, binning, img = st
    P = d.num_surfels
    T = ((W + 15) // 16) * ((Hh + 15) // 16)
    ints = dict(radii=radii.cpu().numpy(), num_rendered=R,
                point_list=_C.read_state("point_list", {}, geom, binning, img, P, W, Hh, torch.int32, R).numpy().view(np.uint32),
                ranges=_C.read_state("ranges", {}, geom, binning, img, P, W, Hh, torch.int32, 2 * T).numpy().view(np.uint32).reshape(-1, 2),
                n_contrib=_C.read_state("n_contrib", {}, geom, binning, img, P, W, Hh, torch.int32, 2 * W * Hh).numpy().view(np.uint32).reshape(2, Hh, W))
    color2, radii2, allmap = dsr.GaussianRasterizer(rs)(means3D=leaves[0], means2D=m2d, opacities=leaves[1], shs=leaves[4],
                                                        scales=leaves[2], rotations=leaves[3])
    torch.autograd.backward([color2, allmap], [dc, do])
    grads = dict(dL_dmeans3D=leaves[0].grad, dL_dopacity=leaves[1].grad, dL_dscales=leaves[2].grad,
                 dL_drotations=leaves[3].grad, dL_dsh=leaves[4].grad, dL_dmeans2D=m2d.grad)
    return color2.detach(), allmap.detach(), grads, ints


def budget_consumed(report, previous_path):
    """Per configuration / build / pair: the largest fraction of a frozen budget (tests/ref_budgets.py) any tensor consumes
    now -- outlier fraction over allowed fraction, worst error over allowed worst error -- next to the same number of an
    earlier report.  (Budgets floor at 1e-4 / 5e-2, so small fractions are the rule.)"""
    from tests.ref_budgets import BUDGET
    prev = {}
    if previous_path and os.path.exists(previous_path):
        prev = json.load(open(previous_path))
    out = {}

    def consumed(rep, name, budget_name):
        res = {}
        for variant in ("strict", "default"):
            if not isinstance(rep.get(name, {}).get(variant), dict):
                continue
            for pair in ("oracle_vs_ref", "product_vs_ref"):
                fl = rep[name][variant][pair]["floats"]
                bud = BUDGET.get(budget_name, {}).get(variant, {}).get(pair)
                if not bud:
                    continue
                worst_t, worst_v = None, -1.0
                for t, (bf, bw) in bud.items():
                    if t not in fl:
                        continue
                    v = max(fl[t]["outlier_frac"] / bf, (fl[t]["worst_rel"] / bw) if fl[t]["outlier_frac"] > 0 else 0.0)
                    if v > worst_v:
                        worst_t, worst_v = t, v
                res[f"{variant}/{pair}"] = {"max_fraction_of_budget": round(worst_v, 4), "tensor": worst_t}
        return res

    for name in report:
        if name.startswith("_"):
            continue
        budget_name = name
        now = consumed(report, name, budget_name)
        before = consumed(prev, name, budget_name) if name in prev else {}
        out[name] = {k: dict(v, previous=before.get(k, {}).get("max_fraction_of_budget")) for k, v in now.items()}
        print("budget", name, json.dumps(out[name]), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="tiny,ragged,small,deg1,subpixel,huge,init_opacity,mid,cfgA,cfgB,cfgE_slice,cfgE_full")
    ap.add_argument("--previous", default=os.path.join(ROOT, "profiles", "r02_ref_parity.json"),
                    help="an earlier report to print next to this one")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_parity.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    so.set_threads(min(64, os.cpu_count() or 1))
    report = {}
    for name in args.configs.split(","):
        kw = CONFIGS[name]
        sc = make_scene(**kw)
        W, Hh, P = sc.width, sc.height, sc.num_surfels
        T = ((W + 15) // 16) * ((Hh + 15) // 16)
        t0 = time.perf_counter()
        st = oracle_forward(sc)
        dc, do = make_upstream_grads(W, Hh)
        og = so.backward(st, dc, do)
        t_or = time.perf_counter() - t0
        d = sc.to(dev)
        dcg, dog = dc.to(dev), do.to(dev)
        pc, po, pg, pints = run_product(d, dcg, dog, W, Hh)
        entry = {"surfels": P, "width": W, "height": Hh, "num_rendered_oracle": int(st["num_rendered"]),
                 "oracle_seconds": round(t_or, 2)}
        for variant in ("strict", "default"):
            if not ref.available(variant):
                entry[variant] = "library not built"
                continue
            ref.use(variant)
            rf = ref.forward(d)
            rg = ref.backward(d, rf, dcg, dog)
            R = int(rf["num_rendered"])
            rints = dict(radii=to_np(rf["radii"]), num_rendered=R, tiles_touched=ref.state("tiles_touched", P),
                         point_list=ref.state("point_list", R), sorted_keys=ref.state("sorted_keys", R),
                         ranges=ref.state("ranges", 2 * T).reshape(-1, 2),
                         n_contrib=ref.state("n_contrib", 2 * W * Hh).reshape(2, Hh, W))
            oracle_i = {"radii": int_diff(st["radii"], rints["radii"]),
                        "radius_max_delta": int(np.abs(st["radii"].astype(np.int64) - rints["radii"]).max()),
                        "num_rendered": [int(st["num_rendered"]), R],
                        "tiles_touched": int_diff(st["tiles_touched"], rints["tiles_touched"]),
                        "point_list": int_diff(st["point_list"], rints["point_list"]),
                        "sorted_keys": int_diff(st["point_list_keys"], rints["sorted_keys"]),
                        "ranges": int_diff(st["ranges"], rints["ranges"]),
                        "n_contrib": n_contrib_diff(st["n_contrib"], rints["n_contrib"]),
                        "n_contrib_total": int(2 * W * Hh)}
            vis = rints["radii"] > 0
            oracle_i["transMat"] = float_stats(st["transMat"][vis], ref.state("transMat", P * 9).reshape(P, 9)[vis], rtol=1e-6)
            oracle_f = compare_images(st["color"], st["others"], og, rf, rg)
            prod_i = {"radii": int_diff(pints["radii"], rints["radii"]),
                      "num_rendered": [int(pints["num_rendered"]), R],
                      "point_list": int_diff(pints["point_list"], rints["point_list"]),
                      "ranges": int_diff(pints["ranges"], rints["ranges"]),
                      "n_contrib": n_contrib_diff(pints["n_contrib"], rints["n_contrib"])}
            prod_f = compare_images(pc, po, pg, rf, rg)
            entry[variant] = {"oracle_vs_ref": {"integers": oracle_i, "floats": oracle_f},
                              "product_vs_ref": {"integers": prod_i, "floats": prod_f}}
        entry["product_vs_oracle"] = {
            "integers": {"radii": int_diff(pints["radii"], st["radii"]),
                         "point_list": int_diff(pints["point_list"], st["point_list"]),
                         "ranges": int_diff(pints["ranges"], st["ranges"]),
                         "n_contrib": int_diff(pints["n_contrib"], st["n_contrib"])},
            "floats": compare_images(pc, po, pg, {"color": st["color"], "others": st["others"]}, og)}
        report[name] = entry
        print(name, json.dumps(entry)[:2000], flush=True)
    report["_budget_consumed"] = budget_consumed(report, args.previous)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(report, open(args.out, "w"), indent=1)
    print("written", args.out)


if __name__ == "__main__":
    main()
