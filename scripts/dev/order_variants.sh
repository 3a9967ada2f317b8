#!/bin/bash
# Development aid (GPU box): launch order / lanes of the forward buckets.  bash scripts/dev/order_variants.sh
set -o pipefail
run() { local label=$1; shift; echo "== $label $*"; env "$@" bash scripts/dev/bench_quick.sh "" | tail -1; }
run default A=1 &&
run b0_second ULTRA_DEV_FWD_ORDER=4,0,3,2,1 &&
run b0_first ULTRA_DEV_FWD_ORDER=0,4,3,2,1 &&
run b0_second_p56 ULTRA_DEV_FWD_ORDER=4,0,3,2,1 ULTRA_DEV_FWD_LANES=3,0,1,5,6,2,3,7 &&
run b0_second_p46 ULTRA_DEV_FWD_ORDER=4,0,3,2,1 ULTRA_DEV_FWD_LANES=3,0,1,4,6,2,3,7 &&
run order_43021 ULTRA_DEV_FWD_ORDER=4,3,0,2,1
