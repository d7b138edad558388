#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q --timeout=900 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tail -20
timeout 600 python bench.py --steps 100 --warmup 40 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print("bench", round(d["value"]), round(d["repeats"]["median"]), d["stage_ms_avg"])'
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
cp variants/trace.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 600 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c_trace.txt
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
