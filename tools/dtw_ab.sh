#!/bin/bash
# A/B of DTW builds: tools/dtw_ab.sh lib1.so lib2.so ...  ("default" = the in-tree library); env ABN_DTW_WGS passes through
for l in "$@"; do
  if [ "$l" = default ]; then python tools/dtw_time.py 2>&1 | grep "dtw 10000"; else ABNET3_HIP_LIB=$l python tools/dtw_time.py 2>&1 | grep "dtw 10000"; fi || exit 1
done
