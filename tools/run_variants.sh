#!/bin/bash
# On the GPU box: for each variants/*.so, swap it in for the product library and run the bench without its CPU legs.
cd "$(dirname "$0")/.."
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for v in ${@:-variants/*.so}; do
  cp $v vidu4d_amd/csrc/libvidu4d_surfel.so
  for st in ${STACKED:-1}; do
  timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --stacked $st 2>/dev/null | V=$v ST=$st python -c '
import json, os, sys
d = json.loads(sys.stdin.readlines()[-1])
keep = os.environ.get("STAGES", "blend")   # STAGES=all prints every stage timer
print(os.environ["V"], "stacked=" + os.environ["ST"], round(d["value"]), round(d["repeats"]["median"]),
      {k: round(v, 4) for k, v in d["stage_ms_avg"].items() if keep == "all" or keep in k})'
  done
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
