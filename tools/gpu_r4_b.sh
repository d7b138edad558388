#!/bin/bash
# round 4, GPU call B: suite without -x, why are recorded segments slower?  tail probe, segment-length variants, SQ counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/b_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/b_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/b_pytest.log | tail -30
echo "== tail probe, recorded segments"; timeout 300 python tools/tail_probe.py 2>/dev/null | tee gpurun_out/b_tail_rec.txt
echo "== tail probe, whole-tile backward"; VIDU4D_SURFEL_WHOLE_TILE_BWD=1 timeout 300 python tools/tail_probe.py 2>/dev/null | tee gpurun_out/b_tail_whole.txt
echo "== variants"; STAGES=all timeout 900 bash tools/run_variants.sh variants/rec*.so 2>&1 | tee gpurun_out/b_variants.txt
echo "== SQ counters recorded"; timeout 400 python tools/pmc_kernel.py blend_bwd --groups 0 --out gpurun_out/b_pmc_rec.json 2>&1 | tail -12
echo "== SQ counters whole"; VIDU4D_SURFEL_WHOLE_TILE_BWD=1 timeout 400 python tools/pmc_kernel.py blend_bwd --groups 0 --out gpurun_out/b_pmc_whole.json 2>&1 | tail -12
