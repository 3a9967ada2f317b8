#!/bin/bash
# One round's profile evidence, on the GPU box from the repo root:  bash scripts/profile_round.sh r01
# Kernel times (--kernel-trace --stats), HBM traffic (separate --pmc passes, FETCH_SIZE / WRITE_SIZE) and the
# SQ instruction/occupancy counters of the same bench command.  GPU_MAX_HW_QUEUES is exported here because the
# profiler initialises HIP before bench.py could set it: 13 queues, bench.py's own default, clamped to the
# pool's limit of 32 and exported under both names (ultra_amd/hwq.py: the bench line then records the count in force).
set -o pipefail
tag=${1:-r01}
q=${ULTRA_HW_QUEUES:-13}; [ "$q" -ge 1 ] 2>/dev/null || q=13; [ "$q" -gt 32 ] && q=32
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=$q ULTRA_HW_QUEUES=$q
out=gpurun_out/prof_$tag
mkdir -p $out
B="python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0 --pipe-leg 0"
S="python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0 --pipe-leg 0"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- $B > $out/kt.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- $S > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- $S > $out/write.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/sq -o sq --output-format csv -- $S > $out/sq.log 2>&1 || exit 1
python3 scripts/timeline.py $out/kt/kt_kernel_trace.csv > $out/timeline.txt
ls $out/*
