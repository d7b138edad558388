#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/u; mkdir -p $O
bash tools/gpu_r4_q.sh "$@"
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
cp variants/$1.so vidu4d_amd/csrc/libvidu4d_surfel.so
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -x > $O/pytest_$1.log 2>&1; tail -3 $O/pytest_$1.log
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
