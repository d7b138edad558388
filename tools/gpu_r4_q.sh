#!/bin/bash
# A/B on one box: product vs the variants named on the command line (variants/<name>.so), twice round-robin
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=d["stage_ms_avg"]; print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]), "fwd", s["blend_fwd"], "bwd", s["blend_bwd"], "sum", round(sum(s.values()),4))'
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for i in 1 2; do
  for v in product "$@"; do
    if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
    timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" $v
  done
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
