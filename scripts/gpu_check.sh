#!/bin/bash
# GPU round trip used during development: parity tests, then a short bench line (no CPU leg).
# usage (on the GPU box, from the repo root): bash scripts/gpu_check.sh <tag> [extra bench args]
tag=${1:-dev}; shift
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --cpu-seconds 0 "$@" > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log > gpurun_out/bench_$tag.json
python scripts/bench_summary.py gpurun_out/bench_$tag.json
