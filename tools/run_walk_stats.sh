#!/bin/bash
# Tile-walk statistics of blend_bwd (bench.py's counting pass) on the results-table scenes.
cd "$(dirname "$0")/.."
walk() {
  python -c '
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print(sys.argv[1], "|", json.dumps(d["roofline"]["limiter"]["tile_walk"]))' "$1"
}
B="--cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0"
timeout 600 python bench.py $B 2>/dev/null | walk "ball"
timeout 600 python bench.py $B --scene object --object-radius 1.0 2>/dev/null | walk "object r=1.0"
timeout 600 python bench.py $B --scene object --object-radius 0.3 2>/dev/null | walk "object r=0.3"
timeout 600 python bench.py $B --surfels 1000000 --res 1920 --height 1080 --frames 24 --steps 20 --warmup 10 2>/dev/null | walk "1M 1080p"
