#!/bin/bash
# builds variants of ONE translation unit (-D flags) into tools/variants/lib_<name>.so
#   tools/variants.sh tower "" "-DFEXP_NODMA" "-DFEXP_NOMFMA -DFEXP_NOFRAG"
# run one with ABNET3_HIP_LIB=tools/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
unit=$1; shift
mkdir -p tools/variants
extra=""; [ "$unit" = dtw ] && extra="-ffp-contract=off"
for v in "$@"; do
  name=${v//-D/}; name=${name// /_}; [ -z "$name" ] && name=base
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC $extra $v -c abnet3_amd/csrc/$unit.hip -o tools/variants/${unit}_$name.o &
done
wait
for v in "$@"; do
  name=${v//-D/}; name=${name// /_}; [ -z "$name" ] && name=base
  objs=$(ls abnet3_amd/lib/obj/*.o | grep -v "/$unit.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/lib_$name.so $objs tools/variants/${unit}_$name.o
  echo built lib_$name.so
done
