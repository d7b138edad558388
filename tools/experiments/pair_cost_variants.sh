cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for V in ${VS:-product fwd6}; do
  if [ $V = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$V.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
  echo "== $V"; python tools/experiments/pair_cost.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
