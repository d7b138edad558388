"""Condenses `rocprofv3 --kernel-trace --stats -- python tools/fit_profile.py` (full Stage-3 step) into a per-step table.
Usage: python tools/fit_kernel_stats.py <..._kernel_stats.csv> <steps traced> > profiles/<tag>_fit_step_kernel_stats.csv"""
import csv, sys

path, steps = sys.argv[1], float(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if float(r["Calls"]) / steps >= 0.1]  # (one-off initialisation kernels dropped)
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"# rocprofv3 --kernel-trace --stats of tools/fit_profile.py (full Stage-3 step: 2 frames, 200k surfels, 512^2, dense ball),"
      f" per optimizer step over {int(steps)} steps (6 of them warm-up); serialised kernel time {total:.0f} us per step")
print("kernel,calls_per_step,us_per_step,avg_us")
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    name = r["Name"].replace(",", " ")[:100]
    print(f"{name},{float(r['Calls']) / steps:.2f},{float(r['TotalDurationNs']) / steps / 1e3:.1f},{float(r['AverageNs']) / 1e3:.1f}")
