#!/bin/bash
# occupancy probe of the whole-tile forward: unused dynamic LDS caps the workgroups per CU (22 KiB static each)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=d["stage_ms_avg"]; print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]), "fwd", s["blend_fwd"], "bwd", s["blend_bwd"])'
for i in 1 2; do for pad in 0 5000 9000 14000 20000; do
VIDU4D_FWD_PAD_LDS=$pad timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 3 --per-frame-surface 0 2>/dev/null | python -c "$P" pad=$pad
done; done
