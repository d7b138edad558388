#!/bin/bash
# The configurations of BASELINE.md section 5, one bench.py run each (GPU box): value / median / algorithmic GB/s.
cd "$(dirname "$0")/.."
run() {
  local tag="$1"; shift
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 5 --per-frame-surface 0 "$@" 2>/dev/null | TAG="$tag" python -c '
import json, os, sys
d = json.loads(sys.stdin.readlines()[-1])
print(os.environ["TAG"], "|", round(d["value"]), "/", round(d["repeats"]["median"]), "| GB/s", round(d["algorithmic_GBps_whole_op"]),
      "| R", round(d["config"]["num_rendered_mean"]), "|", {k: round(v, 4) for k, v in d["stage_ms_avg"].items()})'
}
run "50k 256^2" --surfels 50000 --res 256 --frames 32
run "200k 512^2" 
#run "200k 512^2 --stacked 0" --stacked 0
run "1M 1920x1080" --surfels 1000000 --res 1920 --height 1080 --frames 24 --steps 20 --warmup 10
run "200k 512^2 object r=1.0" --scene object --object-radius 1.0
run "200k 512^2 object r=0.3" --scene object --object-radius 0.3
VIDU4D_BENCH_FORCE_DIST=1 timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | tail -1 > gpurun_out/bench_line_1rank_rccl.json
python -c '
import json
d = json.load(open("gpurun_out/bench_line_1rank_rccl.json")); print("1-rank RCCL |", round(d["value"]), "/", round(d["repeats"]["median"]))'
