#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q --timeout=800 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|Error" | tail -10
echo "== contention"; timeout 600 python tools/contention_probe.py 2>/dev/null > gpurun_out/r04_contention.json; python -c "
import json; d=json.load(open('gpurun_out/r04_contention.json')); print(d['backward_slowdown'], d['blend_bwd_slowdown']); print({k:v['blend_bwd'] for k,v in d['runs'].items()})"
echo "== fetch calib"; timeout 600 bash tools/fetch_calib.sh 2>&1 | tee gpurun_out/r04_fetch_calib.txt
echo "== cfgE"; timeout 1500 python bench.py --surfels 1000000 --res 1920 --height 1080 --frames 240 --cpu-images 2 --torch-cpu-images 0 --fit-densify-steps 0 > gpurun_out/j_cfgE.log 2>&1; tail -c 400 gpurun_out/j_cfgE.log
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/j_cfgE.log") if l.startswith("{")][-1])
json.dump(d, open("gpurun_out/r04_bench_line_cfgE.json","w"), indent=1)
print("cfgE", round(d["value"],1), d["stage_ms_avg"], "fit", d.get("fit_step",{}).get("images_per_s"), d.get("fit_step_geometry",{}).get("images_per_s"), "per-frame", d.get("value_per_frame_calls",{}).get("value"))
PY
