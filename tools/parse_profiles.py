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
         "preprocess_fwd": "preprocess_fwd", "surfel_color": "surfel_color", "preprocess_bwd": "preprocess_bwd",
         "tile_sort_kernel": "tile_sort", "tile_scan": "tile_scan", "tile_order": "tile_scan",
         "blend_seg_T": "blend_seg_T", "blend_combine": "blend_combine"}
# FETCH_SIZE counts every 128-byte read request as 64 bytes: x2 for every kernel.  The guide (MI355X_MICROARCH.md, HBM)
# establishes this for wide coalesced streaming reads and calls other patterns uncalibrated; tools/ubench/fetch_calib.hip
# (profiles/r04_fetch_calib.txt) measured the blend kernels' own pattern -- a gather of records with seven / eight 16-byte
# loads per lane: 128-byte-aligned records read exactly once report 1/2 of their bytes like the stream does; the product's
# 112-byte records report ~1.0x their bytes, i.e. 2x counted = the ~1.9 LINES a record straddling line boundaries pulls.
# (Round 3 used 1.0 for the blend kernels and said so: its traffic figure was a lower bound.)
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
        corr = 2.0
        traffic[k] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "fetch_correction": corr,
                      "hbm_bytes_per_launch": (fe * corr + wr) * 1024.0}
        f.write(f"surfel::{k},{fe:.1f},{wr:.1f},{len(d['FETCH_SIZE'])}\n")
json.dump(traffic, open(os.path.join(summ, "pmc_traffic.json"), "w"), indent=1)

# 3. SQ counters: per-kernel mean per launch
sq = find("SQ", "*counter_collection.csv")
if sq:
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sq)):
        name = r.get("Kernel_Name", "")
        if "surfel::" not in name:
            continue
        key = next((v for k, v in SHORT.items() if k in name), None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
    # average duration per kernel from the un-instrumented trace pass
    dur = defaultdict(list)
    if stats:
        for r in csv.DictReader(open(stats)):
            key = next((v for k, v in SHORT.items() if k in r["Name"]), None)
            if key and "surfel::" in r["Name"]:
                dur[key].append((float(r["AverageNs"]), int(r["Calls"])))
    SIMDS, CLK_GHZ = 1024, 2.4  # 256 CU x 4 SIMD; peak engine clock: a wave64 VALU instruction occupies its SIMD for 4 cycles
    with open(os.path.join(summ, f"{tag}_pmc_sq.csv"), "w") as f:
        f.write("# valu_issue_util = SQ_INSTS_VALU * 4 cycles / (1024 SIMDs * avg duration * 2.4 GHz): share of the VALU issue slots used\n")
        f.write("kernel,launches," + ",".join(c + "_per_launch" for c in cols) + ",avg_duration_us,valu_issue_util\n")
        for k, d in acc.items():
            m = {c: (sum(d[c]) / len(d[c]) if d[c] else float("nan")) for c in cols}
            n = len(next(iter(d.values())))
            # (tile_scan aggregates three kernels: no single duration)
            ns = max(dur[k], key=lambda t: t[1])[0] if dur.get(k) and k != "tile_scan" else float("nan")
            util = m["SQ_INSTS_VALU"] * 4.0 / (SIMDS * ns * CLK_GHZ) if ns == ns else float("nan")
            f.write(f"surfel::{k},{n}," + ",".join(f"{m[c]:.4g}" for c in cols) + f",{ns / 1e3:.1f},{util:.3f}\n")
            if k in traffic and ns == ns:
                # Share of the VALU issue capacity the kernel's wave64 VALU instructions take, bracketed: a plain
                # instruction issues every 2.4 cycles per SIMD on this chip (tools/ubench/valu_rate.hip; DPP adds 6,
                # lane swaps / transcendentals 8), the nominal figure is 4.  Clock: 2.4 GHz peak (SQ_BUSY_CYCLES / 32
                # shader engines / duration gives the sustained one, ~2.26 GHz under these kernels).
                clk = m["SQ_BUSY_CYCLES"] / 32.0 / ns if m["SQ_BUSY_CYCLES"] == m["SQ_BUSY_CYCLES"] else float("nan")
                use_clk = clk if clk == clk and clk > 0.5 else CLK_GHZ
                # HEADLINE: the guide's issue rate -- a wave64 VALU instruction takes 2 cycles on a SIMD-32
                # (MI355X_MICROARCH.md: execution model, per-instruction table) -- at the clock the kernel sustained
                guide = m["SQ_INSTS_VALU"] * 2.0 / (SIMDS * ns * use_clk)
                lo = m["SQ_INSTS_VALU"] * 2.4 / (SIMDS * ns * CLK_GHZ)
                hbm = traffic[k]["hbm_bytes_per_launch"] / (ns * 1e-9) / 8.0e12
                wc = m["SQ_WAVE_CYCLES"]
                traffic[k]["limiter"] = {"bound": "valu" if guide > 0.35 and hbm < 0.3 else ("hbm" if hbm >= 0.3 else "latency"),
                                         "valu_issue_frac": round(guide, 3),
                                         "valu_issue_frac_basis": "SQ_INSTS_VALU x 2 cycles (guide: wave64 on a SIMD-32) / (1024 SIMDs x duration x sustained clock)",
                                         "wait_any_frac_of_wave_cycles": round(m["SQ_WAIT_ANY"] / wc, 3) if wc else None,
                                         "wait_inst_any_frac_of_wave_cycles": round(m["SQ_WAIT_INST_ANY"] / wc, 3) if wc else None,
                                         "mean_resident_waves_per_simd": round(wc * 4.0 / (SIMDS * ns * use_clk), 2) if wc else None,
                                         "note_measured_issue_interval": {"valu_issue_frac_at_2.4_cycles_2.4GHz": round(lo, 3),
                                                                          "source": "tools/ubench/valu_rate.hip: a plain wave64 VALU instruction every 2.4 cycles, DPP 6, lane swaps / transcendentals 8"},
                                         "sustained_clock_ghz": round(clk, 2), "valu_insts_per_launch": m["SQ_INSTS_VALU"],
                                         "counter_hbm_frac_of_8TBps": round(hbm, 4), "avg_duration_us": round(ns / 1e3, 1),
                                         "source": f"rocprofv3 --pmc SQ_ACTIVE_INST_VALU ... ({tag}_pmc_sq.csv)"}
    json.dump(traffic, open(os.path.join(summ, "pmc_traffic.json"), "w"), indent=1)

# 4. L2 (TCC) counters: per-kernel mean per launch, hit rate (round 5: SURVEY.md 8d asks for an L2 / MALL-level figure)
tcc = find("TCC", "*counter_collection.csv")
if tcc:
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(tcc)):
        name = r.get("Kernel_Name", "")
        if "surfel::" not in name:
            continue
        key = next((v for k, v in SHORT.items() if k in name), None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    if stats:
        for r in csv.DictReader(open(stats)):
            key = next((v for k, v in SHORT.items() if k in r["Name"]), None)
            if key and "surfel::" in r["Name"] and key != "tile_scan":
                if key not in dur or int(r["Calls"]) > dur[key][1]:
                    dur[key] = (float(r["AverageNs"]), int(r["Calls"]))
    with open(os.path.join(summ, f"{tag}_pmc_tcc.csv"), "w") as f:
        f.write("# per launch; L2 request = one 128-byte line (TCC_REQ_sum); TCC_EA0_RDREQ_sum: read requests that left the L2 towards "
                "the fabric (Infinity Cache + HBM; FETCH_SIZE = this x 64 B)\n")
        f.write("kernel,launches,TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum,TCC_EA0_RDREQ_sum,l2_hit_rate,l2_request_GBps_at_128B\n")
        for k, d in acc.items():
            m = {c: (sum(v) / len(v)) for c, v in d.items() if v}
            hit, miss, req = m.get("TCC_HIT_sum", float("nan")), m.get("TCC_MISS_sum", float("nan")), m.get("TCC_REQ_sum", float("nan"))
            rate = hit / (hit + miss) if hit + miss > 0 else float("nan")
            ns = dur.get(k, (float("nan"), 0))[0]
            bw = req * 128.0 / ns if ns == ns else float("nan")   # bytes per ns = GB/s
            n = len(next(iter(d.values())))
            f.write(f"surfel::{k},{n},{hit:.4g},{miss:.4g},{req:.4g},{m.get('TCC_EA0_RDREQ_sum', float('nan')):.4g},{rate:.4f},{bw:.1f}\n")
            if k in traffic:
                traffic[k]["l2"] = {"hit_rate": round(rate, 4), "requests_per_launch": req, "l2_request_GBps_at_128B_per_request": round(bw, 1),
                                    "fabric_read_requests_per_launch": m.get("TCC_EA0_RDREQ_sum"),
                                    "note": "per-XCD L2 in front of the Infinity Cache; the counters cannot tell Infinity-Cache hits from HBM "
                                            "reads (guide: FETCH_SIZE counts both) -- the step's working set (~100 MB) fits the 256 MiB cache"}
    json.dump(traffic, open(os.path.join(summ, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in traffic.items()}))
