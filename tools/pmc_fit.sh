#!/bin/bash
# SQ counters of the kernels of the full Stage-3 fitting step (tools/fit_profile.py); prints per-launch means for the kernels
# whose name contains $1 (default: lbs_skin|skin_field).  GPU box, through gpurun.
PAT=${1:-"lbs_skin|skin_field"}
R=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_fit; rm -rf $OUT; mkdir -p $OUT
FIT_K=10 FIT_NO_TORCH_PROF=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace -d $OUT -o pmc --output-format csv -- python $R/tools/fit_profile.py > $OUT/log.txt 2>&1
python - "$OUT" "$PAT" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: f"{sum(x) / len(x):.4g}" for c, x in v.items()})
PY
