#!/bin/bash
# round 5, run B: what blend_bwd's pieces cost now (ablation builds: results wrong, timing only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5b; mkdir -p $O; export TMPDIR=/tmp
bash tools/run_variants.sh variants/new.so variants/abl1.so variants/abl2.so variants/abl3.so variants/abl4.so variants/abl7.so variants/new.so 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
