#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/ah; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for c in auto 0; do for regime in 0 8001; do
VIDU4D_SURFEL_CHUNKS=$c FIT_STEP0=$regime FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $O/fit$regime -o trace --output-format csv -- \
    python $R/tools/fit_profile.py > $O/fit$regime.log 2>&1
f=$(find $O/fit$regime -name '*kernel_stats.csv' | head -1)
echo "== chunks=$c regime $regime"; python $R/tools/fit_kernel_stats.py $f 36 | grep "blend_\|tile_sort" | cut -c1-140
rm -rf $O/fit$regime
done; done
