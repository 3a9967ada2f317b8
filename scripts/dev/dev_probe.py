"""Development aid: device-planned batches only (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ultra_amd import synth, index, lib
from ultra_amd import classify_read_with_mams as stage
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
preset = sys.argv[3] if len(sys.argv) > 3 else "ont_cdna"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
w = synth.make_world(seed=20241008, **dict(synth.WORLDS["c3"], n_genes=genes))
fw = synth.FlatWorld(w)
rs = synth.make_reads_native(fw, n, seed=5, preset=preset)
ix = index.NativeIndex(*w.index())
dix = index.DeviceIndex(ix)
ctx = lib.Context(0)
min_acc = synth.PRESETS[preset]["min_acc"]
for rep in range(reps):
    t1 = time.time(); ctx.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
    t2 = time.time(); stage.run_staged(ctx, min_acc)
    t3 = time.time(); res = ctx.mam_fetch(want_names=True)
    t4 = time.time()
    cells, ms = ctx.myers_timing(); mm = ctx.mam_timing()
    print("dev rep %d: plan+stage %.1f ms  run %.1f ms  fetch %.1f ms | cells/read %.2fM pairs %d | myers fwd %.1f rev %.1f tb %.1f all %.1f fwdk %.1f | build %.1f chain %.1f gather %.1f | e2e %.0f reads/s"
          % (rep, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, cells / n / 1e6, ctx.mam_counts()[2], ms[0], ms[1], ms[2], ms[3], ms[4], mm[0], mm[1], mm[2], n / (t4 - t1)), flush=True)
