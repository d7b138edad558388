#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/i_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/i_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/i_pytest.log | tail -30
P='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"]), round(d["repeats"]["median"]) if "repeats" in d else "", d["stage_ms_avg"])'
timeout 600 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 5 --per-frame-surface 0 2>/dev/null | python -c "$P" product
python - <<'PY'
import torch, sys, numpy as np
sys.path.insert(0,'.')
from tests.test_gpu_round4 import _run, GRAD_NAMES
from tests.util import oracle_forward
from oracle import surfel_oracle as so
from vidu4d_amd import _C, _lib
from vidu4d_amd.synthetic import make_scene, make_object_scene, make_upstream_grads
dev=torch.device('cuda:0')
_C._SPLIT="0"
so.set_threads(64)
for name,sc,boost in (("headline", make_scene(200_000,512,512),1.0), ("uniform60k", make_scene(60_000,256,256,seed=31),1.0), ("object_dist x100", make_object_scene(40_000,256,radius=0.3,opacity_mode="init"),100.0), ("uniform60k_dist x100", make_scene(60_000,256,256,seed=31),100.0)):
    dc,do=make_upstream_grads(sc.width,sc.height)
    do=do.clone(); do[6]*=boost
    st=oracle_forward(sc); g=so.backward(st,dc,do)
    dcd,dod=dc.to(dev),do.to(dev)
    w1=_run(sc,dev,dcd,dod,flags=_lib.DEBUG_WHOLE_TILE_BACKWARD)
    r1=_run(sc,dev,dcd,dod,flags=0)
    def err(a,ref): return {k: float(np.abs(a["grads"][k].cpu().numpy()-ref[k]).max()/(np.abs(ref[k]).max()+1e-30)) for k in GRAD_NAMES if k in ref}
    def err2(a,b): return {k: float((a["grads"][k]-b["grads"][k]).abs().max())/(float(b["grads"][k].abs().max())+1e-30) for k in GRAD_NAMES}
    print(name, "recorded vs whole :", {k:f"{v:.1e}" for k,v in err2(r1,w1).items()})
    print(name, "whole vs oracle   :", {k:f"{v:.1e}" for k,v in err(w1,g).items()})
    print(name, "recorded vs oracle:", {k:f"{v:.1e}" for k,v in err(r1,g).items()})
PY
