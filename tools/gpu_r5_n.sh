#!/bin/bash
# round 5: the fitting step with networks that train -- targeted tests, host probe, profiler tables (GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5n; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dense_stack.py tests/test_gpu_lbs.py tests/test_gpu_stage3.py tests/test_gpu_distributed.py tests/test_gpu_train_dataset.py -q --timeout=600 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert|capture failed" $O/pytest.log | tail -12
python tools/fit_host_probe_optim_warp.py > $O/fit_host_ow.txt 2>&1; grep "FIT_HOST\|capture failed" $O/fit_host_ow.txt
python tools/fit_optim_warp_profile.py > $O/prof.txt 2>&1; grep "OPTIM_WARP\|capture failed" $O/prof.txt
