#!/bin/bash
# round 4, the numbers the documents quote, from the final tree (GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 1500 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log | tail -1 > $O/r04_bench_line.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1; grep "^{" $O/bench_driver_form.log | tail -1 > $O/r04_bench_line_driver_form.json
python - <<'PY'
import json
for f in ("r04_bench_line.json","r04_bench_line_driver_form.json"):
    d=json.load(open("gpurun_out/final/"+f))
    print(f, round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"], "roofline", round(d["roofline"]["frac"],4), d["roofline"]["traffic"])
    for k in ("fit_step","fit_step_geometry","fit_step_densify"):
        v=d.get(k,{}); print("  ",k, v.get("images_per_s"), v.get("ms_per_step"), v.get("surfels_after"))
    print("   per_frame", d.get("value_per_frame_calls",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_pytorch",{}).get("value"))
PY
bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -1 $O/profile_round.log; cp gpurun_out/prof_r04/summary/* $O/
bash tools/profile_fit.sh r04 > /dev/null 2>&1; cp gpurun_out/fit_r04/r04_*.csv gpurun_out/fit_r04/r04_fit_ab.txt $O/; cat $O/r04_fit_ab.txt | head -5
timeout 600 python tools/recorded_precision.py 2>/dev/null | tee $O/r04_recorded_precision.txt
timeout 900 python tools/fuzz_footprint_gpu.py 200 0 > $O/fuzz_a.txt 2>&1; timeout 600 python tools/fuzz_footprint_gpu.py 24 5 large > $O/fuzz_b.txt 2>&1
grep -hv amdgpu.ids $O/fuzz_a.txt $O/fuzz_b.txt > $O/r04_fuzz_footprint_gpu.txt; tail -4 $O/r04_fuzz_footprint_gpu.txt | cut -c1-400
timeout 1200 python tools/ref_parity_report.py --out $O/r04_ref_parity.json > $O/refparity.log 2>&1; grep "^budget cfgE_full\|^budget cfgB" $O/refparity.log | cut -c1-300
timeout 300 python tools/tail_probe.py 2>/dev/null | tee $O/r04_tail_probe.txt
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/trace.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 600 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids > $O/r04_bwd_trace.txt; cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
head -3 $O/r04_bwd_trace.txt | cut -c1-200
bash tools/results_table.sh 2>&1 | tee $O/r04_results_table.txt | cut -c1-120
timeout 600 python bench.py --replicas 1 --fit-steps 30 2>/dev/null | tail -1 > $O/r04_bench_line_replicas1.json; cut -c1-300 $O/r04_bench_line_replicas1.json
cp gpurun_out/bench_line_1rank_rccl.json $O/r04_bench_line_1rank_rccl.json 2>/dev/null
