#!/usr/bin/env python
"""Where the wall time of a step goes on the GPU's side: from a `rocprofv3 --kernel-trace` CSV, the kernels of the last N steps
(a step = the span between two launches of the named anchor kernel), per step: launches, summed kernel time, busy time (union
of the kernels' intervals: branches overlap), idle time between kernels, and the idle gaps by size.
Usage: python tools/trace_gaps.py <kernel_trace.csv> <anchor kernel substring> [steps]"""
import csv, sys
from collections import Counter

path, anchor = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
TABLE = "--table" in sys.argv
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
marks = [i for i, r in enumerate(rows) if anchor in r[2]]
marks = marks[-(steps + 1):]
tot = Counter()
gaps = Counter()
names = Counter()
calls = Counter()
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    span = rows[b][0] - seg[0][0]
    busy, end = 0, seg[0][0]
    for s, e, n in seg:
        if s > end:
            g = s - end
            gaps["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"] += g
            busy += e - s
            end = e
        elif e > end:
            busy += e - end
            end = e
        names[n[:70]] += e - s
        calls[n[:70]] += 1
    if rows[b][0] > end:
        gaps["tail"] += rows[b][0] - end
    tot["launches"] += len(seg)
    tot["kernel_sum_ns"] += sum(e - s for s, e, _ in seg)
    tot["busy_ns"] += busy
    tot["span_ns"] += span
n = max(1, len(marks) - 1)
print(f"{n} steps: launches per step {tot['launches'] / n:.0f}, span {tot['span_ns'] / n / 1e3:.0f} us, summed kernel time "
      f"{tot['kernel_sum_ns'] / n / 1e3:.0f} us, GPU busy (union) {tot['busy_ns'] / n / 1e3:.0f} us, idle {(tot['span_ns'] - tot['busy_ns']) / n / 1e3:.0f} us")
print("idle time per step by gap size (us):", {k: round(v / n / 1e3, 1) for k, v in sorted(gaps.items())})
print("largest kernels (us per step):", [(k, round(v / n / 1e3, 1)) for k, v in names.most_common(12)])
if TABLE:
    print("kernel, calls per step, us per step")
    for k, v in names.most_common():
        print(f"{k.replace(',', ' ')}, {calls[k] / n:.2f}, {v / n / 1e3:.1f}")
