#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5m; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest.log | tail -12
for r in 1.0 0.7 0.5; do for regime in 0 8001; do
echo "== radius $r step0 $regime, forced split: position order (8) vs ordered (0), forward AND backward"; VIDU4D_SURFEL_SPLIT=1 python tools/split_order_ab.py $r $regime 8 0 2>&1 | grep "^flags"
done; done | tee $O/ab2.txt
