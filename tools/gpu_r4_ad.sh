#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/trace.so vidu4d_amd/csrc/libvidu4d_surfel.so
TRACE_SPLIT=0 TRACE_OBJECT_RADIUS=1.0 TRACE_SIGMA_PX=6 timeout 600 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-330
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
bash tools/gpu_r4_aa.sh 2>&1 | grep "blend_\|FIT_STEP"
