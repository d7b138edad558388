#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/d_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/d_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/d_pytest.log | tail -30
timeout 1200 python bench.py > gpurun_out/d_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/d_bench.log") if l.startswith("{")][-1])
print("bench", round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"])
for k in ("fit_step","fit_step_geometry","fit_step_densify"):
    v=d.get(k,{}); print(k, v.get("images_per_s"), v.get("ms_per_step"), v.get("surfels_before"), v.get("surfels_after"), v.get("densification_events"))
print("per_frame", d.get("value_per_frame_calls",{}).get("value"))
PY
