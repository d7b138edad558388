#!/bin/bash
# ONE parametrised runner for everything that is measured on the MI355X box through gpurun (it replaces the 63 one-off
# tools/gpu_r4_*.sh / gpu_r5_*.sh scripts of rounds 4-5).  Usage, from the repository root:
#     gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <task> [args...] ; bash tools/gpu_run.sh <task> ...'
# Every task writes under gpurun_out/<tag>/ (merged back into the build container); the summaries that are to be judged are
# copied into profiles/ by hand afterwards.  TAG=<name> in the environment changes the output directory (default r06).
#
#   tests [pytest args]      the GPU suite (default: tests -m gpu -q)
#   bench <name> [args]      one bench.py line -> <name>.json  (the default command when no args are given)
#   opbench <name> [args]    bench.py's op-level figure only (no fit steps / CPU legs / per-frame surface): quick A/B lines
#   xcd_ab                   item 1(i): the XCD-local schedule, block sizes 0 1 2 4 8, op-level + the TCC (L2) counters
#   profile <tag> [args]     tools/profile_round.sh (kernel stats + FETCH / WRITE / SQ / TCC passes)
#   fit_ab                   the fitting step's regimes (tools/fit_profile.py) with the schedule / graph switches
#   pair_ab                  item 2: two workgroups for the longest tiles of the whole-tile forward, A/B at the headline and on the fit scenes
#   pair_evidence            item 2: the A/B files behind DESIGN.md's paired-workgroup numbers
#   matrix                   the final-tree measurement matrix: cfgA / cfgB (headline, default + driver form) / cfgE lines
#   py <script> [args]       any tools/*.py under a timeout, output kept
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${TAG:-r06}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
TASK=${1:-tests}; shift || true
QUICK="--cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --per-frame-surface 0 --host-probe 0 --fit-optim-warp 0 --fit-steps 0"

line() {  # prints the fields of a bench line that the A/B tables quote
python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[1], "no JSON line:", e); sys.exit(0)
st = d.get("stage_ms_avg", {})
print(sys.argv[1].split("/")[-1], "value", round(d["value"], 1), "median", d.get("repeats", {}).get("median"),
      "blend_fwd", st.get("blend_fwd"), "blend_bwd", st.get("blend_bwd"), "roofline", round(d["roofline"]["frac"], 4))
for k in ("fit_step", "fit_step_geometry", "fit_step_densify", "fit_step_optim_warp", "fit_step_optim_warp_unfused",
          "fit_step_optim_warp_eager", "fit_step_captured"):
    v = d.get(k)
    if v: print("   ", k, v.get("images_per_s"), v.get("ms_per_step"))
pf = d.get("value_per_frame_calls")
if pf: print("    per_frame", pf.get("value"), pf.get("single_stream", {}).get("value"), "host", d.get("host_cost"))
PY
}

case "$TASK" in
tests)
    if [ $# -eq 0 ]; then set -- tests -m gpu -q; fi
    timeout 2400 python -m pytest "$@" --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
    grep -E "^(FAILED|ERROR)|passed|failed|rc " $O/pytest.log | tail -15
    ;;
bench)
    NAME=$1; shift
    timeout 1500 python bench.py "$@" > $O/$NAME.log 2>&1; grep "^{" $O/$NAME.log | tail -1 > $O/$NAME.json; line $O/$NAME.json
    ;;
opbench)
    NAME=$1; shift
    timeout 600 python bench.py $QUICK --repeats 3 "$@" > $O/$NAME.log 2>&1; grep "^{" $O/$NAME.log | tail -1 > $O/$NAME.json; line $O/$NAME.json
    ;;
xcd_ab)
    for B in 0 1 2 4 8 0; do
        VIDU4D_SURFEL_XCD_BLOCK=$B timeout 600 python bench.py $QUICK --repeats 3 "$@" > $O/xcd_$B.log 2>&1
        grep "^{" $O/xcd_$B.log | tail -1 > $O/xcd_$B.json; echo -n "B=$B "; line $O/xcd_$B.json
    done | tee $O/r06_xcd_schedule_ab.txt
    R=$(pwd); cd /tmp
    for B in 0 2 4 8; do
        VIDU4D_SURFEL_XCD_BLOCK=$B rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/$O/tcc_$B -o pmc \
            --output-format csv -- python $R/bench.py --steps 4 --warmup 2 $QUICK --repeats 0 --no-stage-timers "$@" > $R/$O/tcc_$B.log 2>&1
        python $R/tools/gpu_run_counters.py $R/$O/tcc_$B blend_bwd blend_fwd | sed "s/^/B=$B /"
    done | tee -a $R/$O/r06_xcd_schedule_ab.txt
    cd $R
    ;;
profile)
    T=${1:-r06}; shift || true
    bash tools/profile_round.sh $T "$@" > $O/profile_round_$T.log 2>&1; tail -3 $O/profile_round_$T.log
    mkdir -p $O/summary_$T; cp gpurun_out/prof_$T/summary/* $O/summary_$T/ 2>/dev/null
    ;;
fit_ab)
    timeout 1200 python tools/fit_profile.py "$@" 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | tee $O/fit_ab.txt | tail -30
    ;;
graph_env_ab)
    # the captured fitting step against the eager loop, and what the HIP runtime's graph switches do to it
    {
    for W_ in 0 1; do
      for E in "FIT_OPTS={\"captured_step\":false}" "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_GRAPH_BATCH_SIZE=512"; do
        echo -n "optim_warp=$W_ $E : "
        if [ $W_ = 1 ]; then X1="FIT_OPTIM_WARP=1 FIT_STEP0=12001"; else X1="FIT_STEP0=0"; fi
        env $X1 "$E" FIT_K=80 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep FIT_STEP
      done
    done
    } | tee $O/r06_graph_env_ab.txt
    R=$(pwd); cd /tmp
    for C in true false; do
      FIT_OPTS="{\"captured_step\":$C}" FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $R/$O/fittrace_$C -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/$O/fittrace_$C.log 2>&1
      f=$(find $R/$O/fittrace_$C -name '*kernel_stats.csv' | head -1)
      python $R/tools/fit_kernel_stats.py $f 36 > $R/$O/r06_fit_step_kernel_stats_captured_$C.csv; head -3 $R/$O/r06_fit_step_kernel_stats_captured_$C.csv; grep FIT_STEP $R/$O/fittrace_$C.log
      rm -rf $R/$O/fittrace_$C
    done
    cd $R
    ;;
trace_gaps)
    # GPU-side anatomy of the fitting step with training networks: captured graph against the eager loop
    R=$(pwd); cd /tmp
    for C in true false; do
      FIT_OPTIM_WARP=1 FIT_STEP0=12001 FIT_OPTS="{\"captured_step\":$C}" FIT_K=40 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace -d $R/$O/gaps_$C -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/$O/gaps_$C.log 2>&1
      f=$(find $R/$O/gaps_$C -name '*kernel_trace.csv' | head -1)
      echo "captured_step=$C $(grep FIT_STEP $R/$O/gaps_$C.log)"; python $R/tools/trace_gaps.py $f blend_bwd_kernel 20
      python $R/tools/trace_gaps.py $f blend_bwd_kernel 20 --table > $R/$O/r06_fit_optim_warp_kernels_captured_$C.txt
      rm -rf $R/$O/gaps_$C
    done | tee $R/$O/r06_fit_optim_warp_gaps.txt
    cd $R
    ;;
ubench)
    # a microbenchmark of tools/ubench (built here if the binary did not travel): ubench <name>
    N=$1; [ -x tools/ubench/$N ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o tools/ubench/$N tools/ubench/$N.hip
    timeout 600 tools/ubench/$N | tee $O/r06_ubench_$N.txt
    ;;
seg_variants)
    # the segment-parallel path with 256-entry segments (a variant build) against the product's 512, on the fitting scene
    cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
    {
    for V in product seg256; do
      if [ $V = seg256 ]; then cp variants/seg256.so vidu4d_amd/csrc/libvidu4d_surfel.so; export VIDU4D_SEG_LEN=256; else cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; unset VIDU4D_SEG_LEN; fi
      for SPLIT in auto 1; do for SG in 0 1; do
        echo -n "$V split=$SPLIT spec_geom=$SG : "
        VIDU4D_SURFEL_SPLIT=$SPLIT VIDU4D_SURFEL_SPEC_GEOM=$SG timeout 600 python bench.py $QUICK --fit-steps 60 --repeats 0 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'fit', round(d['fit_step']['images_per_s']), round(d['fit_step']['ms_per_step'],3), 'geometry', round(d['fit_step_geometry']['images_per_s']), round(d['fit_step_geometry']['ms_per_step'],3))"
      done; done
    done
    } | tee $O/r06_seg256_ab.txt
    cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
    ;;
pair_ab)
    # item 2: paired workgroups for the longest tiles of the whole-tile forward (VIDU4D_SURFEL_PAIR=K: the tiles above K / 4 x
    # the mean list length) against one workgroup per tile: headline op level (full instance), the bench's fitting scene
    # (colour + alpha / planes 0-4 instances), tools/fit_profile.py's dense ball.
    # VARIANTS="product name ..." swaps variants/<name>.so in (tools/make_variants.sh); KS="0 4 8"; PROFILE=1 adds the kernel times.
    cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
    for V in ${VARIANTS:-product}; do
    if [ $V = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$V.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
    for K in ${KS:-0 4 6 8 0}; do
      echo -n "$V PAIR=$K bench: "
      VIDU4D_SURFEL_PAIR=$K timeout 600 python bench.py $QUICK --fit-steps ${FIT_STEPS:-100} --repeats 3 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_avg',{}); print(round(d['value']), 'median', round(d.get('repeats',{}).get('median')), 'blend_fwd', st.get('blend_fwd'), 'blend_bwd', st.get('blend_bwd'), '| fit', round(d['fit_step']['images_per_s']), round(d['fit_step']['ms_per_step'],3), 'geometry', round(d['fit_step_geometry']['images_per_s']), round(d['fit_step_geometry']['ms_per_step'],3))"
      for S0 in 0 8001; do
        echo -n "$V PAIR=$K dense ball step0=$S0: "
        VIDU4D_SURFEL_PAIR=$K FIT_STEP0=$S0 FIT_K=100 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep FIT_STEP | sed "s/.*step: //"
      done
      if [ "$PROFILE" = 1 ]; then
        R=$(pwd); cd /tmp
        for S0 in 0 8001; do
          VIDU4D_SURFEL_PAIR=$K FIT_STEP0=$S0 FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $R/$O/f8trace -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/$O/f8trace.log 2>&1
          f=$(find $R/$O/f8trace -name '*kernel_stats.csv' | head -1)
          echo "$V PAIR=$K dense ball step0=$S0 kernels: $(python $R/tools/fit_kernel_stats.py $f 36 | grep 'blend_fwd\|blend_bwd' | sed 's/(int.*,\([0-9.]*\)$/ \1/; s/void surfel:://' | tr '\n' ' ')"
          rm -rf $R/$O/f8trace
        done
        cd $R
      fi
    done; done | tee $O/r06_pair_ab.txt
    cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
    ;;
pair_evidence)
    # item 2, the files behind DESIGN.md's numbers: kernel times of bench.py's own fitting scene, whole tiles + pairs against
    # the segment-parallel forward over object radii, what a pair costs per blend instance, the forward's workgroup trace
    KS="0 4 6 8" bash tools/experiments/pair_bench_fit_kernels.sh 2>&1 | tee $O/r06_pair_bench_fit_kernels.txt
    bash tools/experiments/pair_vs_split.sh 2>&1 | tee $O/r06_pair_vs_split.txt
    python tools/experiments/pair_cost.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_pair_cost.txt
    [ -f variants/ftrace.so ] && { bash tools/experiments/pair_trace2.sh; KS="0 6" bash tools/experiments/pair_trace.sh; } 2>&1 | tee $O/r06_pair_fwd_trace.txt
    bash tools/experiments/pair_saturating.sh 2>&1 | tee $O/r06_pair_saturating.txt
    ;;
matrix)
    timeout 1500 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log | tail -1 > $O/r06_bench_line_cfgB.json; line $O/r06_bench_line_cfgB.json
    timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; grep "^{" $O/bench_driver.log | tail -1 > $O/r06_bench_line_cfgB_driver_form.json; line $O/r06_bench_line_cfgB_driver_form.json
    timeout 900 python bench.py --surfels 50000 --res 256 --frames 32 --cpu-images 0 --torch-cpu-images 0 > $O/bench_cfgA.log 2>&1; grep "^{" $O/bench_cfgA.log | tail -1 > $O/r06_bench_line_cfgA.json; line $O/r06_bench_line_cfgA.json
    timeout 1500 python bench.py --surfels 1000000 --res 1920 --height 1080 --frames 240 --cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --host-probe 0 > $O/bench_cfgE.log 2>&1; grep "^{" $O/bench_cfgE.log | tail -1 > $O/r06_bench_line_cfgE.json; line $O/r06_bench_line_cfgE.json
    ;;
py)
    S=$1; shift
    timeout 1500 python $S "$@" 2>&1 | grep -v "amdgpu.ids" | tee $O/$(basename $S .py).txt | tail -40
    ;;
*)
    echo "unknown task $TASK"; exit 2
    ;;
esac
