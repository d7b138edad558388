#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5h; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do for env in "VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=2.0" "VIDU4D_SURFEL_SPLIT_AUTO_TILES_PER_CU=3.5" "VIDU4D_SURFEL_SPLIT=1"; do
env $env timeout 900 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 60 --fit-densify-steps 0 --fit-optim-warp 0 --host-probe 0 --repeats 0 --per-frame-surface 0 --steps 20 --no-stage-timers 2>/dev/null | tail -1 | E=$env python -c '
import json,sys,os
d=json.loads(sys.stdin.read())
print(os.environ["E"], {k:(round(d[k]["images_per_s"]), round(d[k]["ms_per_step"],3)) for k in ("fit_step","fit_step_geometry")})'
done; done | tee $O/fit3.txt
