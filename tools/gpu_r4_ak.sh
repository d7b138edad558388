#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out/ak; mkdir -p $O; export TMPDIR=/tmp
bash tools/gpu_r4_q.sh "$@"
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for v in product "$@"; do
if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
cd /tmp
for regime in 0 8001; do
FIT_STEP0=$regime FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $O/fit$regime -o trace --output-format csv -- python $R/tools/fit_profile.py > $O/fit$regime.log 2>&1
f=$(find $O/fit$regime -name '*kernel_stats.csv' | head -1)
echo "== $v regime $regime: $(grep FIT_STEP $O/fit$regime.log | cut -c40-90)"; python $R/tools/fit_kernel_stats.py $f 36 | grep "blend_" | cut -c14-48,100-140
rm -rf $O/fit$regime
done
cd $R
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
