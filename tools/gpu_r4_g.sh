#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/g_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/g_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/g_pytest.log | tail -30
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]) if "repeats" in d else "", d["stage_ms_avg"], "per-frame-surface", d.get("value_per_frame_calls",{}).get("value"))'
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 5 2>/dev/null | python -c "$P" stacked
