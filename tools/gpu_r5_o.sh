#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5o; mkdir -p $O; export TMPDIR=/tmp
timeout 800 python tools/fit_optim_warp_ab.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tee $O/r05_fit_optim_warp_ab.txt | tail -12
