#!/bin/bash
# blend.hip built with extra flags, one variant per argument ("name:flags").
#   tools/blend_variants.sh build "prio200:-DSURFEL_BWD_PRIO=200" ...   (here: variants/blend_<name>.so)
#   tools/blend_variants.sh run prio200 ...                                (GPU box: headline bench per variant)
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -I include $flags -c vidu4d_amd/csrc/blend.hip -o /tmp/blend_v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/blend_$name.so $(ls vidu4d_amd/csrc/*.o | grep -v "/blend.o") /tmp/blend_v.o
    echo "built variants/blend_$name.so ($flags)"
  done
  exit 0
fi
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
for name in "$@"; do
  [ "$name" = product ] || cp variants/blend_$name.so vidu4d_amd/csrc/libvidu4d_surfel.so
  timeout 300 python bench.py --cpu-images 0 --torch-cpu-images 0 --repeats 3 --per-frame-surface 0 ${BENCH_ARGS:-} 2>/dev/null | N=$name python -c '
import json, os, sys
d = json.loads(sys.stdin.readlines()[-1])
print(os.environ["N"], "|", round(d["value"]), "/", round(d["repeats"]["median"]), "| fit", round(d.get("fit_step", {}).get("images_per_s", 0)), round(d.get("fit_step_geometry", {}).get("images_per_s", 0)),
      "| fwd", round(d["stage_ms_avg"]["blend_fwd"], 4), "bwd", round(d["stage_ms_avg"]["blend_bwd"], 4))'
  cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
done
