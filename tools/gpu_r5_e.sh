#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5e; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|Error" $O/pytest.log | tail -15
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 20 --fit-densify-steps 0 --repeats 2 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c '
import json,sys
d=json.loads(sys.stdin.read()); print(round(d["value"]), d["repeats"], d["stage_ms_avg"], d["value_per_frame_calls"]["value"], d["fit_step"].get("images_per_s"), d["fit_step_geometry"].get("images_per_s"), d["host_enqueue_ms_per_step"])'
