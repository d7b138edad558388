cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so; cp variants/btrace.so vidu4d_amd/csrc/libvidu4d_surfel.so
VIDU4D_SURFEL_SPLIT=0 TRACE_OBJECT_RADIUS=1.0 TRACE_SIGMA_PX=6 timeout 300 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
