#!/bin/bash
# verification of the half-walk tree: GPU suite, bench, fuzz (new criteria), recorded precision
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/t; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 900 python bench.py --cpu-images 0 --torch-cpu-images 0 > $O/bench.log 2>&1; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/t/bench_line.json"))
print(round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"], "roofline", round(d["roofline"]["frac"],4))
for k in ("fit_step","fit_step_geometry","fit_step_densify"):
    v=d.get(k,{}); print("  ",k, v.get("images_per_s"), v.get("ms_per_step"))
print("   per_frame", d.get("value_per_frame_calls",{}).get("value"))
print(json.dumps(d["roofline"]["limiter"].get("tile_walk")))
PY
timeout 600 python tools/recorded_precision.py 2>/dev/null | tee $O/r04_recorded_precision.txt | cut -c1-300
timeout 900 python tools/fuzz_footprint_gpu.py 200 0 > $O/fuzz_a.txt 2>&1; timeout 600 python tools/fuzz_footprint_gpu.py 24 5 large > $O/fuzz_b.txt 2>&1
grep -hv amdgpu.ids $O/fuzz_a.txt $O/fuzz_b.txt > $O/r04_fuzz_footprint_gpu.txt; cut -c1-700 $O/r04_fuzz_footprint_gpu.txt
