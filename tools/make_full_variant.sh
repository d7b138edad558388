#!/bin/bash
# variants/<name>.so = the whole product library compiled with extra flags (for constants that live in shared headers).
#   tools/make_full_variant.sh seg256 "-DSURFEL_SEG_LEN=256 -DSURFEL_SPLIT_MIN=256"
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p variants/obj_$name
C=vidu4d_amd/csrc
for f in preprocess binning blend quaternion lbs bone_tables dense_stack knn post optim skin_field loss contract capi; do
  extra=""; [ $f = blend ] && extra="-fno-slp-vectorize"; [ $f = lbs ] && extra="-Wno-pass-failed"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -I include $extra $flags \
      -c $C/$f.hip -o variants/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so variants/obj_$name/*.o
rm -rf variants/obj_$name
echo "built variants/$name.so ($flags)"
