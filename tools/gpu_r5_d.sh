#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5d; mkdir -p $O; export TMPDIR=/tmp
for r in 0.7 0.85; do for regime in 0; do
  for env in "VIDU4D_SURFEL_SPLIT=auto" "VIDU4D_SURFEL_SPLIT=1" "VIDU4D_SURFEL_SPLIT=auto"; do
  echo -n "[$env] radius $r step0=$regime: "; env $env FIT_STEP0=$regime FIT_PRINT_HINTS=1 FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py $r 2>&1 | grep "FIT_STEP\|FIT_HINTS\|FIT_DEC" | cut -c1-400 | tr '\n' ' '; echo
  done
done; done | tee $O/split_rule2.txt
