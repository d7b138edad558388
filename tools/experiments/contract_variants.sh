cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for V in ${VS:-product c1buf c2x}; do
  if [ $V = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$V.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
  TAG=cv bash tools/gpu_run.sh trace_gaps > /dev/null 2>&1; echo "$V: $(grep -i contract gpurun_out/cv/r06_fit_optim_warp_kernels_captured_true.txt | tail -1)"
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
