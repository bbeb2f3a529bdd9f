#!/bin/bash
# diagnostic build of the library with in-kernel s_memtime stamps (never timed)
set -e
cd "$(dirname "$0")/../abnet3_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DABN_STAMPS -o ../lib/libabnet3_hip.so tower.hip loss.hip ops.hip fbank.hip oneshot.hip -x hip -ffp-contract=off dtw.hip 2>&1 | grep -v warning | head -5
