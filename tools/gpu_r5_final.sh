#!/bin/bash
# round 5, the numbers the documents quote, from the final tree (GPU box).  Split in two calls (A: suite + bench + profiles,
# B: parity report, fuzz, tables) so that neither runs into the per-call limit.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
PART=${1:-A}
O=gpurun_out/final5; mkdir -p $O; export TMPDIR=/tmp
if [ "$PART" = "A" ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -10
timeout 1500 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log | tail -1 > $O/r05_bench_line.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1; grep "^{" $O/bench_driver_form.log | tail -1 > $O/r05_bench_line_driver_form.json
python - <<'PY'
import json
for f in ("r05_bench_line.json","r05_bench_line_driver_form.json"):
    d=json.load(open("gpurun_out/final5/"+f))
    print(f, round(d["value"]), d["repeats"]["median"], d["stage_ms_avg"], "roofline", round(d["roofline"]["frac"],4), d["roofline"]["traffic"])
    for k in ("fit_step","fit_step_geometry","fit_step_densify","fit_step_optim_warp","fit_step_optim_warp_unfused"):
        v=d.get(k,{}); print("  ",k, v.get("images_per_s"), v.get("ms_per_step"), v.get("surfels_after"))
    print("   per_frame", d.get("value_per_frame_calls",{}).get("value"), d.get("value_per_frame_calls",{}).get("single_stream",{}).get("value"), "host", d.get("host_cost"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_pytorch",{}).get("value"))
PY
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -1 $O/profile_round.log; cp gpurun_out/prof_r05/summary/* $O/
timeout 600 python bench.py --opacity init --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --host-probe 0 --repeats 3 2>/dev/null | tail -1 > $O/r05_bench_line_opacity_init.json
python -c '
import json; d=json.load(open("gpurun_out/final5/r05_bench_line_opacity_init.json")); print("opacity init", round(d["value"]), d["stage_ms_avg"], d["config"]["num_rendered_mean"], d["roofline"]["limiter"].get("tile_walk",{}).get("lane_utilisation"))'
bash tools/profile_fit.sh r05 > /dev/null 2>&1; cp gpurun_out/fit_r05/r05_*.csv gpurun_out/fit_r05/r05_fit_ab.txt $O/; cat $O/r05_fit_ab.txt | head -5
else
timeout 1200 python tools/ref_parity_report.py --previous profiles/r04_ref_parity.json --out $O/r05_ref_parity.json > $O/refparity.log 2>&1; grep "^budget cfgE_full\|^budget cfgB" $O/refparity.log | cut -c1-300
timeout 600 python tools/recorded_precision.py 2>/dev/null | tee $O/r05_recorded_precision.txt | tail -3
timeout 900 python tools/fuzz_footprint_gpu.py 200 0 > $O/fuzz_a.txt 2>&1; timeout 600 python tools/fuzz_footprint_gpu.py 24 5 large > $O/fuzz_b.txt 2>&1
grep -hv amdgpu.ids $O/fuzz_a.txt $O/fuzz_b.txt > $O/r05_fuzz_footprint_gpu.txt; tail -4 $O/r05_fuzz_footprint_gpu.txt | cut -c1-400
bash tools/results_table.sh 2>&1 | tee $O/r05_results_table.txt | cut -c1-160
cp gpurun_out/bench_line_1rank_rccl.json $O/r05_bench_line_1rank_rccl.json 2>/dev/null
timeout 900 python bench.py --surfels 50000 --res 256 --frames 32 --cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --host-probe 0 2>/dev/null | tail -1 > $O/r05_bench_line_cfgA.json
timeout 1500 python bench.py --surfels 1000000 --res 1920 --height 1080 --frames 240 --cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --fit-optim-warp 0 --host-probe 0 2>/dev/null | tail -1 > $O/r05_bench_line_cfgE.json
python - <<'PY'
import json
for t in ("cfgA","cfgE"):
    d=json.load(open(f"gpurun_out/final5/r05_bench_line_{t}.json"))
    print(t, round(d["value"],1), d["config"]["workload"][:60], d["stage_ms_avg"], "fit", d.get("fit_step",{}).get("images_per_s"), d.get("fit_step_geometry",{}).get("images_per_s"), "per-frame", d.get("value_per_frame_calls",{}).get("value"))
PY
timeout 600 python bench.py --replicas 1 --fit-steps 30 2>/dev/null | tail -1 > $O/r05_bench_line_replicas1.json; cut -c1-200 $O/r05_bench_line_replicas1.json
fi
