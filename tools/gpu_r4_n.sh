#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/n_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/n_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/n_pytest.log | tail -30
timeout 1200 python bench.py --cpu-images 0 --torch-cpu-images 0 > gpurun_out/n_bench.log 2>&1
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/n_bench.log") if l.startswith("{")][-1])
print("bench", round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"])
for k in ("fit_step","fit_step_geometry","fit_step_densify"):
    v=d.get(k,{}); print(k, v.get("images_per_s"), v.get("ms_per_step"))
print("per_frame", d.get("value_per_frame_calls",{}).get("value"))
PY
bash tools/profile_fit.sh r04b > /dev/null 2>&1; head -3 gpurun_out/fit_r04b/r04b_fit_step_kernel_stats.csv | cut -c1-200; grep -E "copyBuffer|gather|Fill|CUDAFunctor_add|index_elementwise" gpurun_out/fit_r04b/r04b_fit_step_kernel_stats.csv | cut -c1-120; cat gpurun_out/fit_r04b/r04b_fit_ab.txt | head -5
