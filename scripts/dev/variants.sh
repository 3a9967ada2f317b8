#!/bin/bash
# Development aid (GPU box): the bench step under library variants built into variants/*.so (compile-time knobs), each
# swapped in for ultra_amd/libultra_hip.so inside the box's scratch copy of the tree.   bash scripts/dev/variants.sh [bench args]
set -o pipefail
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --check 24 --e2e-rounds 0 --front-rounds 0 --micro 0 --pipe-leg 0 $*"
cp ultra_amd/libultra_hip.so /tmp/base.so
for v in /tmp/base.so variants/*.so; do
    cp $v ultra_amd/libultra_hip.so
    echo "== $v"
    timeout -k 10 300 $B > gpurun_out/var.log 2> gpurun_out/var.err || { tail -3 gpurun_out/var.err; cp /tmp/base.so ultra_amd/libultra_hip.so; exit 1; }
    python - <<'PY'
import json
d = json.loads(open("gpurun_out/var.log").read().strip().splitlines()[-1])
print("   ms/step %.2f  " % d["ms_per_step"] + "  ".join("%s %.2f" % (k, v) for k, v in d["stage_ms_per_step"].items()))
PY
done
cp /tmp/base.so ultra_amd/libultra_hip.so
