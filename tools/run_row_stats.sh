#!/bin/bash
# Tile-walk statistics (quadrant walk vs 4x4-block walk) per scene, then SQ counters of blend_bwd in both modes.
cd "$(dirname "$0")/.."
walk() {
  python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
w = d["roofline"]["limiter"]["tile_walk"]
print(sys.argv[1], "| quadrant walk: trips", w["pair_evaluations"], "incl. batch waits", w["wave_trips_incl_batch_waits"], "| block walk:", w["row_walk"])' "$1"
}
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 2>/dev/null | walk "ball"
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --scene object --object-radius 0.3 2>/dev/null | walk "object r=0.3"
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --surfels 1000000 --res 1920 --height 1080 --frames 24 --steps 20 --warmup 10 2>/dev/null | walk "1M 1080p"
for rw in 0 1; do
  echo "== VIDU4D_BWD_ROW_WALK=$rw"
  VIDU4D_BWD_ROW_WALK=$rw python tools/pmc_kernel.py blend_bwd_kernel --groups ${PMC_GROUPS:-0,1} --out gpurun_out/pmc_row_walk_$rw.json
done
