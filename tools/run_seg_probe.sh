cd $GRAFT_REPO_ROOT
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for v in base seg256; do
  cp variants/$v.so vidu4d_amd/csrc/libvidu4d_surfel.so
  for split in auto 1; do
    VIDU4D_SURFEL_SPLIT=$split timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --fit-steps 0 --repeats 2 --per-frame-surface 0 2>/dev/null | V=$v S=$split python -c '
import json, os, sys
d = json.loads(sys.stdin.readlines()[-1])
print(os.environ["V"], "split=" + os.environ["S"], round(d["value"]), {k: round(v, 4) for k, v in d["stage_ms_avg"].items() if "blend" in k or "sort" in k})'
  done
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
