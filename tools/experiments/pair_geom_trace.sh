# the geometry-regime fitting step on the dense ball, with and without paired workgroups: where the step's time goes on the GPU
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for K in 0 6; do
  VIDU4D_SURFEL_PAIR=$K FIT_STEP0=8001 FIT_K=40 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace -d $R/gpurun_out/pgt_$K -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/gpurun_out/pgt_$K.log 2>&1
  f=$(find $R/gpurun_out/pgt_$K -name '*kernel_trace.csv' | head -1)
  echo "PAIR=$K $(grep FIT_STEP $R/gpurun_out/pgt_$K.log)"; python $R/tools/trace_gaps.py $f blend_bwd_kernel 20
  python $R/tools/trace_gaps.py $f blend_bwd_kernel 20 --table | head -14
  rm -rf $R/gpurun_out/pgt_$K
done
