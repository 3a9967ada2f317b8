"""Development aid: do two batches in flight (two host threads, each cycling over its own contexts) beat one?"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq
hwq.export_hw_queues()
import numpy as np
import torch
from ultra_amd import synth, lib, index as native

world = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
w = synth.make_world(seed=20241008, **synth.WORLDS[world])
flat = synth.FlatWorld(w)
sets = [synth.make_reads_native(flat, n, seed=1000 + b, preset="ont_cdna", threads=8) for b in range(4)]
nix = native.NativeIndex(*w.index())
dix = native.DeviceIndex(nix, 0)
LANES = int(os.environ.get("PROBE_LANES", "0"))
ctxs = [lib.Context(0, cfg={"launch_lanes": LANES} if LANES else None) for _ in sets]
for c, rs in zip(ctxs, sets):
    c.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
for c in ctxs:
    c.mam_run(0.6)
torch.cuda.synchronize()
STEPS = 16
t0 = time.perf_counter()
for i in range(STEPS):
    ctxs[i % 4].mam_run(0.6)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / STEPS * 1e3
print("one batch in flight: %.2f ms/step" % one, flush=True)
for nthreads in (2, 3, 4):
    def work(k):
        mine = ctxs[k::nthreads]
        for i in range(STEPS // nthreads):
            mine[i % len(mine)].mam_run(0.6)
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / (STEPS // nthreads * nthreads) * 1e3
    print("%d batches in flight: %.2f ms/step" % (nthreads, per), flush=True)
