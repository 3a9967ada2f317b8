#!/bin/bash
# Development aid (GPU box): one short bench step per argument set, stage times only.  bash scripts/dev/bench_quick.sh "<bench args>" ...
set -o pipefail
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0"
for args in "$@"; do
    echo "== $args"
    timeout -k 10 300 $B $args > gpurun_out/bq.log 2> gpurun_out/bq.err || { tail -3 gpurun_out/bq.err; exit 1; }
    python - gpurun_out/bq.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("   ms/step %.2f  run %.2f  " % (d["ms_per_step"], d.get("kernels_only", {}).get("ms_per_step", 0)) + "  ".join("%s %.2f" % (k, v) for k, v in s.items()))
PY
done
