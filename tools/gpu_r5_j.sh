#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5j; mkdir -p $O; export TMPDIR=/tmp
bash tools/run_variants.sh variants/base.so variants/bw5.so variants/bw7.so variants/bb64.so variants/bb192.so variants/fw6.so variants/base.so 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
timeout 900 python tools/soak_fit.py 1300 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/soak.txt
