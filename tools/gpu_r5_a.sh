#!/bin/bash
# round 5, run A: parity suite on the new blend kernels, then old / new / nomask A/B on one box, then the host's own cost
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5a; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -5
STAGES=all bash tools/run_variants.sh variants/old.so variants/new.so variants/nomask.so variants/old.so variants/new.so 2>&1 | grep -v amdgpu.ids | tee $O/variants.txt
# host cost of a step: the same Python around a scene whose kernels take next to nothing
for st in 1 0; do
timeout 300 python bench.py --surfels 2000 --res 64 --steps 300 --warmup 20 --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 1 --per-frame-surface 0 --no-stage-timers --stacked $st 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); print("host probe stacked", sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "enqueue", round(d["host_enqueue_ms_per_step"],4))' $st | tee -a $O/host.txt
done
