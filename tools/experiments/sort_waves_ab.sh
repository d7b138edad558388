# the in-LDS sort of the lists beyond 1024 entries with 4 / 8 / 16 waves per tile: the fitting step on the dense ball + kernel times
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
R=$(pwd)
for V in ${VS:-product sort8 sort16 product sort8 sort16}; do
  if [ $V = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$V.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
  echo -n "$V: "
  FIT_STEP0=0 FIT_K=100 FIT_NO_TORCH_PROF=1 timeout 300 python tools/fit_profile.py 2>&1 | grep FIT_STEP | sed "s/.*step: //" | tr '\n' ' '
  cd /tmp; export TMPDIR=/tmp
  FIT_STEP0=0 FIT_K=30 FIT_NO_TORCH_PROF=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/swa -o trace --output-format csv -- python $R/tools/fit_profile.py > $R/gpurun_out/swa.log 2>&1
  f=$(find $R/gpurun_out/swa -name '*kernel_stats.csv' | head -1)
  python $R/tools/fit_kernel_stats.py $f 36 | grep "tile_sort" | sed 's/(unsigned.*,\([0-9.]*\)$/ \1/; s/void surfel:://' | tr '\n' ' '; echo
  rm -rf $R/gpurun_out/swa; cd $R
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
