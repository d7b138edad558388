#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench command      -> gpurun_out/prof/trace
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes -> gpurun_out/prof/{fetch,write}
# then tools/parse_profiles.py condenses them into profiles/<tag>_*.csv and profiles/pmc_traffic.json.
# Usage: tools/profile_round.sh <tag> [extra bench.py args]
set -u
TAG=${1:-r01}; shift || true
R=$(pwd); export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- \
  python $R/bench.py --steps 30 --warmup 5 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 "$@" > $OUT/trace_bench.log 2>&1
tail -1 $OUT/trace_bench.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc --output-format csv -- \
    python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers "$@" > $OUT/$c.log 2>&1
done
# SQ counters (8 SQ slots per pass): instruction mix and stall buckets of the blend kernels
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace -d $OUT/SQ -o pmc --output-format csv -- \
  python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers "$@" > $OUT/SQ.log 2>&1
# L2 (TCC) pass: hits / misses / requests per kernel -- the working set of the headline step (~100 MB) sits in the 256 MiB Infinity
# Cache, so the memory-side counters above (FETCH_SIZE = TCC_EA0_RDREQ x 64 B, Infinity-Cache hits included) are not HBM bytes;
# this pass says what the per-XCD L2s absorb in front of them (MI355X_MICROARCH.md: hit rate = TCC_HIT_sum / (HIT + MISS))
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum \
  --kernel-trace -d $OUT/TCC -o pmc --output-format csv -- \
  python $R/bench.py --steps 4 --warmup 2 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 0 --per-frame-surface 0 --no-stage-timers "$@" > $OUT/TCC.log 2>&1
cd $R
python tools/parse_profiles.py $OUT $TAG
