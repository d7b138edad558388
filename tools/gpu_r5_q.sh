#!/bin/bash
# round 5: soak of the Stage-3 loop with networks that train (densify / prune, opacity reset, AdamW starting mid-run)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5q; mkdir -p $O; export TMPDIR=/tmp
SOAK_TRAIN_NETS=1 timeout 1200 python tools/soak_fit.py 900 2>&1 | grep -v "Warn\|warn\|amdgpu.ids\|run_backward" | tee $O/r05_soak_fit_networks_train.txt | cut -c1-400 | tail -8
