cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/ftrace.so vidu4d_amd/csrc/libvidu4d_surfel.so
for K in ${KS:-0 6}; do
  echo "#### object scene radius 1.0, default footprints, initial opacity, colour + alpha instance, PAIR=$K"
  VIDU4D_SURFEL_PAIR=$K VIDU4D_SURFEL_SPLIT=0 TRACE_AUX=alpha TRACE_OBJECT_RADIUS=1.0 TRACE_OPACITY_MODE=init timeout 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
