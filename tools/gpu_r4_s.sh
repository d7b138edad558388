#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/s; mkdir -p $O
bash tools/gpu_r4_q.sh half halfboth halfboth6
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
cp variants/halfboth.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_halfboth.log 2>&1; tail -5 $O/pytest_halfboth.log
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
