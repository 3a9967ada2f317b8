#!/bin/bash
# Kernel trace + per-kernel stats of an arbitrary python command (development aid), from the repo root on the GPU box:
#   bash scripts/profile_cmd.sh <tag> scripts/dev/dev_probe.py 6000 262144
set -o pipefail
tag=${1:-q}; shift
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=13 ULTRA_HW_QUEUES=13
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
grep -v "^W2026\|^I2026" $out/kt.log | tail -8
python3 scripts/timeline.py $out/kt/kt_kernel_trace.csv > $out/timeline.txt 2>/dev/null
rm -f $out/kt/kt_kernel_trace.csv
python3 - $out/kt/kt_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-90s calls %6s  total %10.3f ms  avg %9.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
