#!/bin/bash
# Bench lines beside the default one (other batch sizes and presets), on the GPU box: bash scripts/bench_variants.sh
set -o pipefail
run() { # name, bench args
    local name=$1; shift
    timeout -k 10 400 python bench.py --cpu-seconds 0 "$@" > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err || { tail -3 gpurun_out/bench_$name.err; return 1; }
    tail -1 gpurun_out/bench_$name.log > gpurun_out/bench_$name.json
    python scripts/bench_summary.py gpurun_out/bench_$name.json > gpurun_out/bench_$name.txt; sed -n 1,2p gpurun_out/bench_$name.txt
}
run isoseq_32k --reads 32768 --preset isoseq && run ont_drna_32k --reads 32768 --preset ont_drna && run 512k_reads --reads 524288 --steps 5
