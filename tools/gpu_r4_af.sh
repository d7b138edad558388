#!/bin/bash
# a variant against the product: headline bench and the dense fit step (both regimes), alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/gpu_r4_q.sh "$@"
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for i in 1 2; do for v in product "$@"; do
  if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
  for regime in 0 8001; do echo -n "$v step0=$regime: "; FIT_STEP0=$regime FIT_K=60 FIT_NO_TORCH_PROF=1 python tools/fit_profile.py 2>&1 | grep "FIT_STEP" | cut -c40-120; done
done; done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
