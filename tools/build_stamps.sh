#!/bin/bash
# diagnostic build of the library with in-kernel s_memtime stamps (never timed, never the product library: it goes to
# tools/variants/lib_stamps.so, which the stamp tools load through ABNET3_HIP_LIB)
set -e
cd "$(dirname "$0")/../abnet3_amd/csrc"
mkdir -p ../../tools/variants
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DABN_STAMPS -o ../../tools/variants/lib_stamps.so tower.hip loss.hip ops.hip fbank.hip oneshot.hip -x hip -ffp-contract=off dtw.hip 2>&1 | grep -v warning | head -5
