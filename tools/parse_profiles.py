"""Condenses rocprofv3 output directories (tools/profile_round.sh) into small CSV / JSON summaries
under gpurun_out/<...>/summary (copy them into profiles/ to have them tracked)."""
import csv, glob, json, os, sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
summ = os.path.join(out_dir, "summary")
os.makedirs(summ, exist_ok=True)


def find(sub, pattern):
    hits = glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


# 1. kernel stats: keep our kernels + the top 10 others
stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "surfel::" in r["Name"] or "vidu4d" in r["Name"] or "lbs_kernel" in r["Name"]]
    others = [r for r in rows if r not in keep][:10]
    with open(os.path.join(summ, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in keep + others:
            r = dict(r)
            r["Name"] = r["Name"][:120]
            w.writerow(r)

# 2. PMC passes: per-kernel mean counter value per launch
SHORT = {"blend_bwd_kernel": "blend_bwd", "blend_fwd_kernel": "blend_fwd", "emit_keys": "emit_keys",
         "preprocess_fwd": "preprocess_fwd", "preprocess_bwd": "preprocess_bwd", "tile_sort_kernel": "tile_sort",
         "tile_scan": "tile_scan", "tile_totals": "tile_scan", "group_prefix": "tile_scan", "tile_order": "tile_scan",
         "blend_seg_T": "blend_seg_T", "blend_combine": "blend_combine"}
STREAMING = {"preprocess_fwd", "preprocess_bwd"}  # wide coalesced streaming reads: FETCH_SIZE counts 1/2 (guide, HBM section)
per = defaultdict(lambda: defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    path = find(counter, "*counter_collection.csv")
    if not path:
        continue
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "surfel::" not in name:
            continue
        if r.get("Counter_Name") != counter:
            continue
        key = next((v for k, v in SHORT.items() if k in name), None)
        if key:
            per[key][counter].append(float(r["Counter_Value"]))
traffic = {}
with open(os.path.join(summ, f"{tag}_pmc_fetch_write.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,launches\n")
    for k, d in per.items():
        fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        corr = 2.0 if k in STREAMING else 1.0
        traffic[k] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "fetch_correction": corr,
                      "hbm_bytes_per_launch": (fe * corr + wr) * 1024.0}
        f.write(f"surfel::{k},{fe:.1f},{wr:.1f},{len(d['FETCH_SIZE'])}\n")
json.dump(traffic, open(os.path.join(summ, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in traffic.items()}))
