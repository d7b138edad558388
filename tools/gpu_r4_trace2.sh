#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/trace2; mkdir -p $O
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/trace.so vidu4d_amd/csrc/libvidu4d_surfel.so
for split in 1 0; do
echo "#### TRACE_SPLIT=$split object radius 1.0 sigma_px 6"
TRACE_SPLIT=$split TRACE_OBJECT_RADIUS=1.0 TRACE_SIGMA_PX=6 timeout 600 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
