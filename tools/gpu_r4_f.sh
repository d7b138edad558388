#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/f_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/f_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/f_pytest.log | tail -30
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]) if "repeats" in d else "", d["stage_ms_avg"], "per-frame-surface", d.get("value_per_frame_calls",{}).get("value"))'
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 2>/dev/null | python -c "$P" stacked
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --stacked 0 --frame-streams 0 --per-frame-surface 0 2>/dev/null | python -c "$P" per-frame-serial
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --stacked 0 --frame-streams 1 --per-frame-surface 0 2>/dev/null | python -c "$P" per-frame-streams
VIDU4D_SURFEL_WHOLE_TILE_BWD=1 timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --stacked 0 --frame-streams 1 --per-frame-surface 0 2>/dev/null | python -c "$P" per-frame-streams-whole
