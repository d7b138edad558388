#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for r in 1.0 0.7 0.5 0.3; do for regime in 0 8001; do
  for env in "VIDU4D_SURFEL_SPLIT_AUTO_LEN=2048" "VIDU4D_SURFEL_SPLIT=0"; do
  echo -n "[$env] radius $r step0=$regime: "; env $env FIT_STEP0=$regime FIT_PRINT_HINTS=1 FIT_K=40 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py $r 2>&1 | grep "FIT_STEP\|FIT_HINTS" | cut -c1-300 | tr '\n' ' '; echo
  done
done; done
