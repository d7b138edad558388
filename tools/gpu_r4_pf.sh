cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for i in 1 2; do for v in product halfwalk; do
if [ $v = product ]; then cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so; else cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so; fi
timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(sys.argv[1], round(d['value']), 'per-frame', round(d['value_per_frame_calls']['value']), d['value_per_frame_calls'].get('ms_per_step'))" $v
done; done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
