#!/bin/bash
# Quick GPU check after a blend-kernel change: rasterizer parity tests, then the bench without its CPU legs.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py -x -q -m gpu 2>&1 | tail -4
for st in 1 0; do
  timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --stacked $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('stacked=$st', round(d['value']), d['repeats']['median'], {k:round(v,4) for k,v in d['stage_ms_avg'].items()})"
done
