#!/bin/bash
# Long-list sort mode (Vidu4dSurfelForwardArgs::long_list_sort) per scene: forced MSD split, forced one-workgroup, hinted default.
cd "$(dirname "$0")/.."
line() {
  python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], "|", round(d["value"]), "/", round(d["repeats"]["median"]), "| fit_step", round(d.get("fit_step", {}).get("images_per_s", 0)), round(d.get("fit_step_geometry", {}).get("images_per_s", 0)),
      "| sort ms", round(d["stage_ms_avg"].get("tile_sort", 0), 4), "| longest", d["config"].get("longest_tile_list"))' "$1"
}
for from in 0 2147483648 10000; do
  export VIDU4D_MSD_SORT_FROM=$from
  echo "== MSD_SORT_FROM=$from"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --repeats 3 --per-frame-surface 0 2>/dev/null | line "200k 512^2 ball (+fit)"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --scene object --object-radius 1.0 2>/dev/null | line "object r=1.0"
  timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 --scene object --object-radius 0.3 2>/dev/null | line "object r=0.3"
done
