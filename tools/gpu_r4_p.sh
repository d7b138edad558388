#!/bin/bash
# half walk A/B (one box): product vs variants/half.so (6 waves, 5 spilled registers) vs variants/half5.so (5 waves)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/p; mkdir -p $O; export TMPDIR=/tmp
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=d["stage_ms_avg"]; print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]), "fwd", s["blend_fwd"], "bwd", s["blend_bwd"], "sum", round(sum(s.values()),4))'
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for i in 1 2; do
  for v in product half half5; do
    if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
    timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" $v
  done
done
cp variants/half.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_half.log 2>&1; tail -5 $O/pytest_half.log
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 600 python tools/recorded_precision.py 2>/dev/null | tee $O/r04_recorded_precision.txt
timeout 900 python tools/fuzz_footprint_gpu.py 200 0 > $O/fuzz_a.txt 2>&1; timeout 600 python tools/fuzz_footprint_gpu.py 24 5 large > $O/fuzz_b.txt 2>&1
grep -hv amdgpu.ids $O/fuzz_a.txt $O/fuzz_b.txt > $O/r04_fuzz_footprint_gpu.txt; cut -c1-600 $O/r04_fuzz_footprint_gpu.txt
