#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5g; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 40 --fit-densify-steps 0 --repeats 1 --per-frame-surface 0 --steps 30 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c '
import json,sys
d=json.loads(sys.stdin.read())
for k in ("fit_step","fit_step_geometry","fit_step_optim_warp","fit_step_optim_warp_unfused"):
    print(k, round(d[k]["images_per_s"]), round(d[k]["ms_per_step"],3))' || tail -20 $O/bench.log
