#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/ftrace.so vidu4d_amd/csrc/libvidu4d_surfel.so
echo "#### headline"; timeout 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
echo "#### dense ball, 6-pixel footprints, initial opacity (lists walked to their ends)"; TRACE_SPLIT=0 TRACE_OBJECT_RADIUS=1.0 TRACE_SIGMA_PX=6 TRACE_OPACITY_MODE=init timeout 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
