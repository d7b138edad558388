#!/bin/bash
# round 5: rocprofv3 kernel trace of the fitting step with networks that train (GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); export TMPDIR=/tmp
OUT=$R/gpurun_out/fit_ow; mkdir -p $OUT
cd /tmp
K=30
FIT_OPTIM_WARP=1 FIT_STEP0=12001 FIT_K=$K FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- \
    python $R/tools/fit_profile.py > $OUT/run.log 2>&1
grep FIT_STEP $OUT/run.log
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python $R/tools/fit_kernel_stats.py $f $((K + 6)) > $OUT/r05_fit_step_optim_warp_kernel_stats.csv
head -40 $OUT/r05_fit_step_optim_warp_kernel_stats.csv | cut -c1-150
