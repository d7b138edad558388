#!/bin/bash
# round 4, GPU call A: the whole GPU suite on the refactored blend kernels, the GPU fuzz, the reference parity report, a bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
timeout 600 python bench.py --steps 100 --warmup 40 > gpurun_out/a_bench.log 2>&1; tail -c 600 gpurun_out/a_bench.log
VIDU4D_SURFEL_WHOLE_TILE_BWD=1 timeout 600 python bench.py --steps 100 --warmup 40 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 > gpurun_out/a_bench_whole.log 2>&1
timeout 900 python tools/fuzz_footprint_gpu.py 200 0 > gpurun_out/a_fuzz.txt 2>&1; tail -3 gpurun_out/a_fuzz.txt
timeout 600 python tools/fuzz_footprint_gpu.py 24 5 large > gpurun_out/a_fuzz_large.txt 2>&1; tail -3 gpurun_out/a_fuzz_large.txt
timeout 1200 python tools/ref_parity_report.py --out gpurun_out/r04_ref_parity.json > gpurun_out/a_refparity.log 2>&1; tail -3 gpurun_out/a_refparity.log
