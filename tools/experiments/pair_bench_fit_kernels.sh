# kernel times of bench.py's OWN fitting scene (fit_step: <.., 1> instances, fit_step_geometry: <.., 2>), with and without pairs
R=$(pwd); cd /tmp; export TMPDIR=/tmp
QUICK="--cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --per-frame-surface 0 --host-probe 0 --fit-optim-warp 0"
for K in ${KS:-0 6}; do
  VIDU4D_SURFEL_PAIR=$K rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pbf_$K -o trace --output-format csv -- python $R/bench.py $QUICK --fit-steps 40 --repeats 0 --steps 3 --warmup 2 --no-stage-timers > $R/gpurun_out/pbf_$K.log 2>&1
  f=$(find $R/gpurun_out/pbf_$K -name '*kernel_stats.csv' | head -1)
  echo "PAIR=$K"; python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'blend_fwd_kernel' in n or 'blend_bwd_kernel' in n: print('  ', n[:58], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1))
"
  rm -rf $R/gpurun_out/pbf_$K
done
