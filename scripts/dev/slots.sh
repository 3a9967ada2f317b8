#!/bin/bash
# Development aid (GPU box): end positions kept per pair in the first pass (ultra_cfg.loc_slots; bench.py --slots).
set -o pipefail
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --check 24 --e2e-rounds 0 --front-rounds 0 --micro 0 --pipe-leg 0"
for v in 0 16 8 12 24; do
  echo "== slots $v"
  timeout -k 10 300 $B --slots $v $* > gpurun_out/var.log 2>gpurun_out/var.err || { tail -3 gpurun_out/var.err; exit 1; }
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/var.log").read().strip().splitlines()[-1])
print("   ms/step %.2f  " % d["ms_per_step"] + "  ".join("%s %.2f" % (k, v) for k, v in d["stage_ms_per_step"].items()))
PY
done
