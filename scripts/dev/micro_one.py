"""Development aid: one Myers micro-benchmark, a few runs (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq
hwq.export_hw_queues()
import numpy as np
import torch  # noqa: F401
from ultra_amd import synth, lib
name = sys.argv[1] if len(sys.argv) > 1 else "M-contained"
ctx = lib.Context(0)
pairs = synth.make_micro_pairs(name, err=0.07, planted=0.5)
ctx.myers_stage(*pairs)
for _ in range(4):
    ctx.myers_run()
    cells, ms = ctx.myers_timing()
    print(name, [round(x, 2) for x in ms])
r = ctx.myers_fetch()
ed = r["ed"]; hit = ed >= 0
print("hits", int(hit.sum()), "ed percentiles", np.percentile(ed[hit], [10, 50, 90, 99, 100]).tolist())
