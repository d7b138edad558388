#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/v; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 900 python bench.py --cpu-images 0 --torch-cpu-images 0 > $O/bench.log 2>&1; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/v/bench_line.json"))
print(round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"], "roofline", round(d["roofline"]["frac"],4))
for k in ("fit_step","fit_step_geometry","fit_step_densify"):
    v=d.get(k,{}); print("  ",k, v.get("images_per_s"), v.get("ms_per_step"))
print("   per_frame", d.get("value_per_frame_calls",{}).get("value"))
PY
bash tools/profile_fit.sh r04 > /dev/null 2>&1; cp gpurun_out/fit_r04/r04_*.csv gpurun_out/fit_r04/r04_fit_ab.txt $O/; head -5 $O/r04_fit_ab.txt
head -8 $O/r04_fit_step_kernel_stats.csv | cut -c1-110; head -9 $O/r04_fit_step_geometry_kernel_stats.csv | cut -c1-110
