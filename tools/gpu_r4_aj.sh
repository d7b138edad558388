#!/bin/bash
# issue priorities by remaining work: GPU suite, then headline + dense fit step against variants/prev.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/aj; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -5
bash tools/gpu_r4_af.sh prev
bash tools/results_table.sh 2>&1 | cut -c1-100
