#!/usr/bin/env python
"""Per-kernel PMC counters of the bench workload (GPU box, through gpurun):

    python tools/pmc_kernel.py <kernel substring> [--bench-args "..."] [--out gpurun_out/pmc_x.json]

Runs `rocprofv3 --pmc <8 SQ counters> --kernel-trace` once per counter group (counters are collected in their
own runs, never together with the API / memory-copy trace domains) on a short `bench.py` run and prints the
mean value per launch of every counter for the kernels whose name contains the substring."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY",
    "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS",
    "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM",
    "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS SQ_ACTIVE_INST_VMEM",
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--bench-args", default="--steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --no-stage-timers --frame-streams 0 --per-frame-surface 0")
    ap.add_argument("--out", default="")
    ap.add_argument("--groups", default="0,1,2")
    ap.add_argument("--script", default="bench.py", help="the workload, relative to the repo root (its arguments: --bench-args)")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    res = defaultdict(dict)
    for gi in [int(x) for x in args.groups.split(",")]:
        d = f"/tmp/pmc_{os.getpid()}_{gi}"
        cmd = ["rocprofv3", "--pmc"] + GROUPS[gi].split() + ["--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                                                              sys.executable, os.path.join(ROOT, args.script)] + args.bench_args.split()
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("no counter output for group", gi, r.stderr[-500:])
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            name = row.get("Kernel_Name", "")
            if args.kernel in name:
                acc[name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, cs in acc.items():
            for c, v in cs.items():
                res[name][c] = sum(v) / len(v)
                res[name]["launches"] = len(v)
    for name, cs in res.items():
        print("==", name)
        for c, v in cs.items():
            print(f"   {c:32s} {v:14.4g}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
