#!/bin/bash
# Kernel trace + timeline of one short bench run (development aid):  bash scripts/profile_quick.sh <tag> [bench args]
set -o pipefail
tag=${1:-q}; shift
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=13 ULTRA_HW_QUEUES=13
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 4 --warmup 2 --cpu-seconds 0 --check 0 --pipe-leg 0 "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python3 scripts/timeline.py $out/kt/kt_kernel_trace.csv ${TIMELINE_MIN:-0.05} > $out/timeline.txt
rm -f $out/kt/kt_kernel_trace.csv
cat $out/timeline.txt
