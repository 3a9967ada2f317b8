#!/bin/bash
# Development aid (GPU box): the bench step with alternative builds of the library (build/variants/*.so, made on the CPU
# side with -D switches).   bash scripts/dev/so_variants.sh [--test] build/variants/a.so ...
set -o pipefail
B="python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0"
test=0; [ "$1" = --test ] && { test=1; shift; }
cp ultra_amd/libultra_hip.so /tmp/libultra_hip_default.so
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("   ms/step %.2f  run %.2f  " % (d["ms_per_step"], d.get("kernels_only", {}).get("ms_per_step", 0)) + "  ".join("%s %.2f" % (k, v) for k, v in s.items()))
PY
}
rc=0
for so in default "$@"; do
    echo "== $so"
    if [ "$so" = default ]; then cp /tmp/libultra_hip_default.so ultra_amd/libultra_hip.so; else cp "$so" ultra_amd/libultra_hip.so; fi
    if [ $test = 1 ] && [ "$so" != default ]; then
        timeout -k 10 600 python -u -m pytest tests/test_gpu_myers.py tests/test_gpu_classify.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/sv_test.log 2>&1 || { tail -15 gpurun_out/sv_test.log; rc=1; break; }
        tail -1 gpurun_out/sv_test.log
    fi
    timeout -k 10 300 $B > gpurun_out/sv.log 2> gpurun_out/sv.err || { tail -3 gpurun_out/sv.err; rc=1; break; }
    show gpurun_out/sv.log
done
cp /tmp/libultra_hip_default.so ultra_amd/libultra_hip.so
exit $rc
