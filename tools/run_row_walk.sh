#!/bin/bash
# Backward walk A/B (VIDU4D_BWD_ROW_WALK): whole-wave quadrant walk against the per-row 4x4 block walk.
cd "$(dirname "$0")/.."
line() {
  python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], "|", round(d["value"]), "/", round(d["repeats"]["median"]), "| fit_step", round(d.get("fit_step", {}).get("images_per_s", 0)), round(d.get("fit_step_geometry", {}).get("images_per_s", 0)),
      "| fwd", round(d["stage_ms_avg"].get("blend_fwd", 0), 4), "bwd", round(d["stage_ms_avg"].get("blend_bwd", 0), 4))' "$1"
}
for rw in ${ROW_WALKS:-0 1}; do
  export VIDU4D_BWD_ROW_WALK=$rw
  echo "== VIDU4D_BWD_ROW_WALK=$rw"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --repeats 3 --per-frame-surface 0 2>/dev/null | line "200k 512^2 ball (+fit)"
  if [ -z "$QUICK" ]; then
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --scene object --object-radius 1.0 2>/dev/null | line "object r=1.0"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --scene object --object-radius 0.3 2>/dev/null | line "object r=0.3"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --surfels 1000000 --res 1920 --height 1080 --frames 24 --steps 20 --warmup 10 2>/dev/null | line "1M 1920x1080"
  fi
done
