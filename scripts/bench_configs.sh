#!/bin/bash
# The bench lines of the round beside BASELINE configs[2] (bench.py's default), on the GPU box:  bash scripts/bench_configs.sh <tag>
#   configs[1]  chr21-like annotation, 100 000 Iso-Seq-like reads (1 % errors, ~2.5 kb)   --preset isoseq --world c2
#   configs[4]  GRCh38-like annotation, dRNA-like reads (12 % errors, long tail)          --preset ont_drna --world c3
set -o pipefail
tag=${1:-r03}
run() { # name, bench args
    local name=$1; shift
    timeout -k 10 500 python bench.py "$@" > gpurun_out/bench_${tag}_$name.log 2> gpurun_out/bench_${tag}_$name.err || { tail -3 gpurun_out/bench_${tag}_$name.err; return 1; }
    tail -1 gpurun_out/bench_${tag}_$name.log > gpurun_out/bench_${tag}_$name.json
    python scripts/bench_summary.py gpurun_out/bench_${tag}_$name.json | sed -n 1,2p
}
run default && \
run isoseq_c2 --preset isoseq --world c2 --reads 100000 --batches 2 --micro 0 --cpu-seconds 5 && \
run ont_drna_c3 --preset ont_drna --world c3 --reads 1048576 --batches 1 --micro 0 --cpu-seconds 5
