#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5o; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dense_stack.py tests/test_gpu_lbs.py tests/test_gpu_stage3.py -q --timeout=600 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert|capture failed" $O/pytest.log | tail -12
timeout 800 python tools/fit_optim_warp_ab.py 2>&1 | grep -v "Warn\|warn\|amdgpu.ids" | tee $O/r05_fit_optim_warp_ab.txt | tail -12
