#!/bin/bash
# Development aid (GPU box): the bench step under other launch orders of the traceback's band classes (ULTRA_DEV_TB_ORDER, csrc/myers.hip).
set -o pipefail
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0 --pipe-leg 0"
for o in default 0,8,7,6,5,4,3,2,1 8,7,6,5,0,4,3,2,1 8,7,6,5,4,3,0,2,1 8,0,7,6,5,4,3,2,1; do
  echo "== $o"
  if [ $o = default ]; then timeout -k 10 300 $B > gpurun_out/var.log 2>gpurun_out/var.err || exit 1
  else ULTRA_DEV_TB_ORDER=$o timeout -k 10 300 $B > gpurun_out/var.log 2>gpurun_out/var.err || exit 1; fi
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/var.log").read().strip().splitlines()[-1])
print("   ms/step %.2f  " % d["ms_per_step"] + "  ".join("%s %.2f" % (k, v) for k, v in d["stage_ms_per_step"].items()))
PY
done
