#!/bin/bash
# BASELINE configs[4] (1 M surfels, 1920x1080) on one GPU: the bench line incl. fit_step / fit_step_geometry
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --surfels 1000000 --res 1920 --height 1080 --frames 240 --cpu-images 2 --torch-cpu-images 0 --fit-densify-steps 0 > gpurun_out/cfgE.log 2>&1; tail -c 300 gpurun_out/cfgE.log
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/cfgE.log") if l.startswith("{")][-1])
json.dump(d, open("gpurun_out/r04_bench_line_cfgE.json","w"), indent=1)
print("cfgE", round(d["value"],1), d["stage_ms_avg"], "fit", d.get("fit_step",{}).get("images_per_s"), d.get("fit_step_geometry",{}).get("images_per_s"), "per-frame", d.get("value_per_frame_calls",{}).get("value"))
PY
