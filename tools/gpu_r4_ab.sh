#!/bin/bash
# kernel stats of the fit step with the segment-parallel forward forced (old rule) -- after the position-search change
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/ab; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for regime in 0 8001; do
VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=100 FIT_STEP0=$regime FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $O/fit$regime -o trace --output-format csv -- \
    python $R/tools/fit_profile.py > $O/fit$regime.log 2>&1
f=$(find $O/fit$regime -name '*kernel_stats.csv' | head -1)
python $R/tools/fit_kernel_stats.py $f 36 | head -8 | cut -c1-130
done
rm -rf $O/fit0 $O/fit8001
