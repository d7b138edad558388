#!/bin/bash
# chunked long lists (segment_split = -1): the tests, the dense fit step with and without, the bench's fit legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/ag; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -x --timeout=600 -k "chunked or recorded" > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
for i in 1 2; do for c in 0 auto; do for regime in 0 8001; do
  echo -n "chunks=$c step0=$regime: "; VIDU4D_SURFEL_CHUNKS=$c FIT_STEP0=$regime FIT_K=100 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep "FIT_STEP\|Error\|error\|fault" | cut -c40-160
done; done; done
for c in 0 auto; do echo -n "bench chunks=$c: "; VIDU4D_SURFEL_CHUNKS=$c timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-densify-steps 0 --repeats 1 --per-frame-surface 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), d['fit_step']['images_per_s'], d['fit_step_geometry']['images_per_s'])"; done
