# GPU-side anatomy of the reference's per-frame surface at the headline size (one stream): kernel table + idle gaps per step
R=$(pwd); cd /tmp; export TMPDIR=/tmp
QUICK="--cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --host-probe 0 --fit-optim-warp 0 --fit-steps 0 --repeats 0 --no-stage-timers"
rocprofv3 --kernel-trace -d $R/gpurun_out/pft -o trace --output-format csv -- python $R/bench.py $QUICK --per-frame-surface 1 --steps 30 --warmup 5 > $R/gpurun_out/pft.log 2>&1
f=$(find $R/gpurun_out/pft -name '*kernel_trace.csv' | head -1)
grep "^{" $R/gpurun_out/pft.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'per_frame', d['value_per_frame_calls'])" | cut -c1-600
python $R/tools/trace_gaps.py $f blend_bwd_kernel 40 | cut -c1-900
python $R/tools/trace_gaps.py $f blend_bwd_kernel 40 --table | head -24 | cut -c1-140
rm -rf $R/gpurun_out/pft
