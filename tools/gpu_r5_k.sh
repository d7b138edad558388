#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5k; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest.log | tail -12
# dense Stage-3 ball, forced split: serial against parallel repairs, both regimes, and the GEOM instance with / without its pre-pass
for r in 1.0 0.7; do for regime in 0 8001; do
  for env in "VIDU4D_SURFEL_SPLIT=1 VIDU4D_SURFEL_SERIAL_REPAIR=1" "VIDU4D_SURFEL_SPLIT=1" "VIDU4D_SURFEL_SPLIT=1 VIDU4D_SURFEL_SPEC_GEOM=1 VIDU4D_SURFEL_SERIAL_REPAIR=1" "VIDU4D_SURFEL_SPLIT=1 VIDU4D_SURFEL_SPEC_GEOM=1" "VIDU4D_SURFEL_SPLIT=0"; do
  echo -n "[$env] radius $r step0=$regime: "; env $env FIT_STEP0=$regime FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py $r 2>&1 | grep "FIT_STEP" | cut -c1-120
  done
done; done | tee $O/repair.txt
