#!/bin/bash
# fit step: the auto split threshold (whole-tile forward + recorded backward vs segment-parallel forward)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do
for env in "" "VIDU4D_SURFEL_SPLIT=0" "VIDU4D_SURFEL_SPLIT_AUTO_LEN=4096" "VIDU4D_SURFEL_SPLIT_AUTO_LEN=8192"; do
  for regime in 0 8001; do
    echo -n "[$env] step0=$regime: "; env $env FIT_STEP0=$regime FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py 2>&1 | grep "FIT_STEP" | cut -c40-200
  done
done
done
for r in 0.3 0.5; do for env in "" "VIDU4D_SURFEL_SPLIT=0"; do echo -n "[$env] radius $r: "; env $env FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py $r 2>&1 | grep "FIT_STEP" | cut -c40-200; done; done
