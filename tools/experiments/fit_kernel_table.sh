# per-step kernel table of the fitting step (tools/fit_profile.py's dense ball), regime by FIT_STEP0
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for S0 in ${S0S:-0 8001}; do
  FIT_STEP0=$S0 FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fkt -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/gpurun_out/fkt.log 2>&1
  f=$(find $R/gpurun_out/fkt -name '*kernel_stats.csv' | head -1)
  python $R/tools/fit_kernel_stats.py $f 36 > $R/gpurun_out/r06_fit_step_kernel_stats_step$S0.csv
  head -28 $R/gpurun_out/r06_fit_step_kernel_stats_step$S0.csv | cut -c1-150
  rm -rf $R/gpurun_out/fkt
done
