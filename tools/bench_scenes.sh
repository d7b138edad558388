#!/bin/bash
# One bench.py line per scene of the results table (GPU box): value / median, fit-step figures, blend stage times.
#   QUICK=1: the headline scene only (with the two fit-step figures)
cd "$(dirname "$0")/.."
line() {
  python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], "|", round(d["value"]), "/", round(d["repeats"]["median"]), "| fit_step", round(d.get("fit_step", {}).get("images_per_s", 0)), round(d.get("fit_step_geometry", {}).get("images_per_s", 0)),
      "| fwd", round(d["stage_ms_avg"].get("blend_fwd", 0), 4), "bwd", round(d["stage_ms_avg"].get("blend_bwd", 0), 4))' "$1"
}
B="--cpu-images 0 --torch-cpu-images 0 --repeats 3 --per-frame-surface 0"
timeout 600 python bench.py $B 2>/dev/null | line "200k 512^2 ball (+fit)"
[ -n "$QUICK" ] && exit 0
timeout 600 python bench.py $B --fit-steps 0 --scene object --object-radius 1.0 2>/dev/null | line "object r=1.0"
timeout 600 python bench.py $B --fit-steps 0 --scene object --object-radius 0.3 2>/dev/null | line "object r=0.3"
timeout 600 python bench.py $B --fit-steps 0 --surfels 1000000 --res 1920 --height 1080 --frames 24 --steps 20 --warmup 10 2>/dev/null | line "1M 1920x1080"
