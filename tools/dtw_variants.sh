#!/bin/bash
# builds dtw.hip variants (-D flags) into tools/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for v in "$@"; do
  name=${v//-D/}; name=${name// /_}; [ -z "$name" ] && name=base
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off $v -c abnet3_amd/csrc/dtw.hip -o tools/variants/dtw_$name.o
  objs=$(ls abnet3_amd/lib/obj/*.o | grep -v dtw.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/lib_$name.so $objs tools/variants/dtw_$name.o
  echo built lib_$name.so
done
