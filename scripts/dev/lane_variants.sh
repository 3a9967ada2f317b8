#!/bin/bash
# Development aid (GPU box): the bench step under different lane assignments of the forward and traceback launches
# (ULTRA_DEV_FWD_LANES / ULTRA_DEV_TB_LANES - nine band classes; the ninth rides on a stream beyond the lanes whatever its entry says -, csrc/myers.hip).   bash scripts/dev/lane_variants.sh tb|fwd
set -o pipefail
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0"
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("   ms/step %.2f  run %.2f  " % (d["ms_per_step"], d.get("kernels_only", {}).get("ms_per_step", 0)) + "  ".join("%s %.2f" % (k, v) for k, v in s.items()))
PY
}
run() { # label, env assignments...
    local label=$1; shift
    echo "== $label $*"
    env "$@" timeout -k 10 300 $B > gpurun_out/lv.log 2> gpurun_out/lv.err || { tail -3 gpurun_out/lv.err; return 1; }
    show gpurun_out/lv.log
}
if [ "$1" = tb ]; then
    run default A=1 && run swap56 ULTRA_DEV_TB_LANES=0,1,2,3,4,6,5,7,0 && run swap45 ULTRA_DEV_TB_LANES=0,1,2,3,5,4,6,7,0 &&
    run flip ULTRA_DEV_TB_LANES=4,5,6,7,0,1,2,3,0 && run c5main ULTRA_DEV_TB_LANES=0,1,2,3,4,7,6,5,0 && run small_on_67 ULTRA_DEV_TB_LANES=0,1,2,3,4,5,6,6,0
else
    run default A=1 && run parts1 ULTRA_DEV_FWD_LANES=1,0,1,5,6,2,3,4 && run parts2 ULTRA_DEV_FWD_LANES=2,0,1,5,6,2,3,4 &&
    run p2_lane4 ULTRA_DEV_FWD_LANES=3,0,1,5,4,2,3,4 && run p1_4_p2_7 ULTRA_DEV_FWD_LANES=3,0,1,4,7,2,3,4 && run b4_main ULTRA_DEV_FWD_LANES=3,0,1,4,5,2,3,7 &&
    run p2_lane7 ULTRA_DEV_FWD_LANES=3,0,1,5,7,2,3,4
fi
