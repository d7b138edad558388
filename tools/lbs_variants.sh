#!/bin/bash
# lbs.hip built with extra flags, one variant per argument ("name:flags").
#   tools/lbs_variants.sh build "occ2:-DLBS_BX_WAVES_PER_EU=2" ...   (here: variants/lbs_<name>.so)
#   tools/lbs_variants.sh run occ2 ...                                  (GPU box: rocprofv3 kernel stats of tools/lbs_bench.py)
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p variants
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-pass-failed -I include $flags -c vidu4d_amd/csrc/lbs.hip -o /tmp/lbs_v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lbs_$name.so $(ls vidu4d_amd/csrc/*.o | grep -v "/lbs.o") /tmp/lbs_v.o
    echo "built variants/lbs_$name.so ($flags)"
  done
  exit 0
fi
R=$(pwd)
cp vidu4d_amd/csrc/libvidu4d_surfel.so /tmp/product.so
export TMPDIR=/tmp
for name in "$@"; do
  cp variants/lbs_$name.so vidu4d_amd/csrc/libvidu4d_surfel.so
  rm -rf /tmp/lv; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/lv -o x --output-format csv -- python $R/tools/lbs_bench.py > /tmp/lv.log 2>&1)
  f=$(find /tmp/lv -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep 'per step' /tmp/lv.log)"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lbs_skin_kernel" in r["Name"]:
        print("   ", r["Name"][31:75], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
cp /tmp/product.so vidu4d_amd/csrc/libvidu4d_surfel.so
