#!/bin/bash
# rocprofv3 kernel trace of the full Stage-3 fitting step (tools/fit_profile.py), per regime, condensed into
# profiles/<tag>_fit_step[_geometry]_kernel_stats.csv; plus wall-clock A/B lines of the regularised regime with the
# normal term inside the loss kernels (default) and on the per-frame render() path it replaced.
# Usage (on the GPU box, through gpurun): tools/profile_fit.sh <tag>
set -u
TAG=${1:-r03}
R=$(pwd); export TMPDIR=/tmp
OUT=$R/gpurun_out/fit_$TAG; mkdir -p $OUT
cd /tmp
K=30
for regime in 0 8001; do
  name=fit_step; [ $regime != 0 ] && name=fit_step_geometry
  FIT_STEP0=$regime FIT_K=$K FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $OUT/$name -o trace --output-format csv -- \
    python $R/tools/fit_profile.py > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  python $R/tools/fit_kernel_stats.py $f $((K + 6)) > $OUT/${TAG}_${name}_kernel_stats.csv
done
{
  echo "# wall clock of tools/fit_profile.py (200k surfels in a ball, 512^2, 2 frames / step), FIT_K=100"
  for regime in 0 8001; do
    echo "## start step $regime, default path"
    FIT_STEP0=$regime FIT_K=100 FIT_NO_TORCH_PROF=1 python $R/tools/fit_profile.py 2>&1 | grep FIT_STEP
  done
  echo "## start step 8001, normal term on the per-frame render() path (fused_normal_loss off: round 2's path)"
  FIT_STEP0=8001 FIT_K=100 FIT_NO_TORCH_PROF=1 FIT_OPTS='{"fused_normal_loss": false}' python $R/tools/fit_profile.py 2>&1 | grep FIT_STEP
  echo "## start step 8001, every extension off (per-frame calls, torch losses, torch post-processing)"
  FIT_STEP0=8001 FIT_K=30 FIT_NO_TORCH_PROF=1 FIT_FUSED_POST=0 FIT_OPTS='{"fused_loss": false, "stacked_frames": false, "fused_warp": false, "canonical_params": false}' python $R/tools/fit_profile.py 2>&1 | grep FIT_STEP
} > $OUT/${TAG}_fit_ab.txt
cd $R
cat $OUT/${TAG}_fit_ab.txt
