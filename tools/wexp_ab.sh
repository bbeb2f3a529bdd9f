#!/bin/bash
# knock-out builds of the weight-gradient launch (tools/variants.sh tower "" "-DWEXP_NOCONVERT" ...): per-launch times, one box
for l in "$@"; do ABNET3_HIP_LIB=tools/variants/lib_$l.so python tools/ab_step.py 2>&1 | tail -1 || exit 1; done
