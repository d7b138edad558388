cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/ftrace.so vidu4d_amd/csrc/libvidu4d_surfel.so
for K in ${KS:-0}; do echo "#### headline PAIR=$K"; VIDU4D_SURFEL_PAIR=$K timeout 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids; done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
