#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]) if "repeats" in d else "", d["stage_ms_avg"])'
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" product
STAGES=all bash tools/run_variants.sh variants/*.so 2>&1 | tee gpurun_out/l_variants.txt
R=$(pwd)
for v in product rec128B; do
  [ $v = product ] || { cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; }
  (cd /tmp && rm -rf /tmp/pm_$v && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pm_$v -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers > /dev/null 2>&1)
  python - $v <<'PY'
import csv, glob, sys
from collections import defaultdict
f = glob.glob(f"/tmp/pm_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "surfel::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v) / 1024, 1) for k, v in acc.items()}, "(FETCH_SIZE MiB per launch, as counted)")
PY
  [ $v = product ] || cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
done
