#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5l; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/pytest.log | tail -12
cd /tmp
i=0
for regime in 0 8001; do
for env in "VIDU4D_SURFEL_SPLIT=1 VIDU4D_SURFEL_POSITION_ORDER=1" "VIDU4D_SURFEL_SPLIT=1" "VIDU4D_SURFEL_SPLIT=0"; do
  i=$((i+1))
  env $env FIT_STEP0=$regime FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $O/t$i -o trace --output-format csv -- python $R/tools/fit_profile.py 1.0 > $O/t$i.log 2>&1
  f=$(find $O/t$i -name '*kernel_stats.csv' | head -1)
  echo "== step0 $regime $env"; python $R/tools/fit_kernel_stats.py $f 36 | grep "blend\|combine\|repair\|seg_T" | cut -c1-60,100-140
done; done | tee $O/kernels.txt
