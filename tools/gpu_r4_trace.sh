#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/trace; mkdir -p $O
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/trace.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 600 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids > $O/r04_bwd_trace.txt; cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
cat $O/r04_bwd_trace.txt | cut -c1-250
