#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5i; mkdir -p $O; export TMPDIR=/tmp
bash tools/run_variants.sh variants/wide0.so variants/wide1.so variants/wide0.so variants/wide1.so 2>&1 | grep -v amdgpu.ids | tee $O/wide.txt
