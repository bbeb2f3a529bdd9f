for h in 128 256; do for l in 64 128 256; do echo "heavy $h light $l"; ABN_WGRAD_WGS_HEAVY=$h ABN_WGRAD_WGS_LIGHT=$l python tools/ab_step.py 2>&1 | tail -1 || exit 1; done; done
