#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5f; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lbs.py tests/test_gpu_round5.py tests/test_gpu_optim.py tests/test_gpu_stage3.py -q --timeout=600 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest.log | tail -25
