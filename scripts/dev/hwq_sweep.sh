#!/bin/bash
# Development aid (GPU box): the bench step of the three presets at several hardware-queue counts, twice each.
#   bash scripts/dev/hwq_sweep.sh "12 13 16 20 24"
set -o pipefail
for rep in ${REPS:-1 2}; do for q in $1; do
for cfg in "" "--preset isoseq --world c2 --reads 100000 --batches 2" "--preset ont_drna --world c3 --reads 262144 --batches 4"; do
    e2e=0; [ -z "$cfg" ] && e2e=2
    timeout -k 10 240 python bench.py --hw-queues $q --cpu-seconds 0 --front-rounds 0 --e2e-rounds $e2e --check 0 --pipe-leg 0 --micro 0 --steps 12 $cfg > gpurun_out/bench_x.log 2>gpurun_out/bench_x.err || exit 1
    tail -1 gpurun_out/bench_x.log | python -c "
import json,sys;j=json.loads(sys.stdin.read());s=j['stage_ms_per_step'];e=j.get('end_to_end') or {}
print('q=$q', ('$cfg'[9:16] or 'cdna   '), 'step %.2f run %.2f fwd %.2f st %.2f tb %.2f mam %.2f ch %.2f' % (j['ms_per_step'], j['kernels_only']['ms_per_step'], s['myers_forward'], s['myers_starts'], s['myers_traceback'], s['mam_build'], s['chain']), 'e2e %.2fM' % (e.get('reads_per_s', 0) / 1e6))"
done; done; done
