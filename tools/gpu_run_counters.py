#!/usr/bin/env python
"""Mean value per launch of every counter of a rocprofv3 --pmc pass, for the kernels whose name contains one of the given
substrings:  python tools/gpu_run_counters.py <rocprofv3 output dir> <kernel substring> [...]   (used by tools/gpu_run.sh).
Prints the L2 hit rate and the bytes behind the L2 when the TCC counters are there (MI355X_MICROARCH.md: hit rate =
TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); TCC_EA0_RDREQ counts 64-byte requests; wide reads count half: x 2)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, subs = sys.argv[1], sys.argv[2:]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        name = row.get("Kernel_Name", "")
        for s in subs:
            if s in name:
                acc[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for s in subs:
        c = {k: sum(v) / len(v) for k, v in acc[s].items()}
        n = max((len(v) for v in acc[s].values()), default=0)
        line = f"{s}: launches {n} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items()))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            line += f" | L2 hit rate {c['TCC_HIT_sum'] / max(1.0, c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}"
        if "TCC_EA0_RDREQ_sum" in c:
            line += f" | read bytes behind L2 (x2 corrected) {2 * 64 * c['TCC_EA0_RDREQ_sum'] / 1e6:.1f} MB"
        print(line)


if __name__ == "__main__":
    main()
