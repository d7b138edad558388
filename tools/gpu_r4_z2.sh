#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/z2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
for r in 1.0 0.85 0.7 0.5; do for regime in 0 8001; do
  for env in "" "VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=100" "VIDU4D_SURFEL_SPLIT=0"; do
  echo -n "[$env] radius $r step0=$regime: "; env $env FIT_STEP0=$regime FIT_PRINT_HINTS=1 FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py $r 2>&1 | grep "FIT_STEP\|FIT_HINTS" | cut -c40-200 | tr '\n' ' '; echo
  done
done; done
