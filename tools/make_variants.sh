#!/bin/bash
# Builds variants/<name>.so = the product library with blend.hip compiled under extra flags.
#   tools/make_variants.sh base: five:-DSURFEL_BWD_WAVES_PER_EU=5 ...
# tools/run_variants.sh (on the GPU box) swaps each one in and runs the quick bench.
set -e
cd "$(dirname "$0")/.."
python -m vidu4d_amd.build > /dev/null
mkdir -p variants
C=vidu4d_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I include \
      -fno-slp-vectorize $flags -c $C/blend.hip -o variants/blend_$name.o
  objs=$(ls $C/*.o | grep -v "/blend.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $objs variants/blend_$name.o
  rm variants/blend_$name.o
  echo "built variants/$name.so ($flags)"
done
