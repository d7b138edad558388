#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fetch calib"; timeout 600 bash tools/fetch_calib.sh 2>&1 | grep -v warning | tee gpurun_out/r04_fetch_calib.txt
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]) if "repeats" in d else "", d["stage_ms_avg"])'
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" product
STAGES=all bash tools/run_variants.sh variants/*.so 2>&1 | tee gpurun_out/k_variants.txt
timeout 900 python bench.py --surfels 1000000 --res 1920 --height 1080 --frames 240 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" cfgE
