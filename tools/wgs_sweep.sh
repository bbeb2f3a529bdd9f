#!/bin/bash
# workgroups per layer of the 128 x 128 weight-gradient launch: "heavy:light" pairs, C2 step time and launch times on one box
for hl in "$@"; do h=${hl%%:*}; l=${hl##*:}; echo "heavy $h light $l"; ABN_WGRAD_WGS_HEAVY=$h ABN_WGRAD_WGS_LIGHT=$l python tools/ab_step.py 2>&1 | tail -1 || exit 1; done
