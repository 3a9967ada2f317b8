"""Development aid: where a batch's wall time goes (native plan / stage / run / fetch) on a C3-shaped workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq
hwq.export_hw_queues()
import numpy as np
from ultra_amd import synth, index, lib
from ultra_amd import classify_read_with_mams as stage
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
preset = sys.argv[3] if len(sys.argv) > 3 else "ont_cdna"
t = time.time(); w = synth.make_world(seed=20241008, **dict(synth.WORLDS["c3"], n_genes=genes)); print("world %.1fs" % (time.time() - t), flush=True)
fw = synth.FlatWorld(w)
t = time.time(); rs = synth.make_reads_native(fw, n, seed=5, preset=preset); print("reads %.1fs" % (time.time() - t), flush=True)
t = time.time(); ix = index.NativeIndex(*w.index()); print("index %.1fs" % (time.time() - t), flush=True)
ctx = lib.Context(0)
for rep in range(3):
    t0 = time.time(); plan = ix.plan_arrays(rs.nam_off, rs.nam_part.reshape(-1, 3), rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
    t1 = time.time(); stage.stage_batch(ctx, plan)
    t2 = time.time(); stage.run_staged(ctx, synth.PRESETS[preset]["min_acc"])
    t3 = time.time(); res = ctx.mam_fetch()
    t4 = time.time()
    cells, ms = ctx.myers_timing(); mm = ctx.mam_timing()
    print("rep %d: plan %.0f ms (%.1f us/read)  stage %.0f ms  run %.1f ms  fetch %.1f ms | pairs/read %.1f cells/read %.2fM | myers fwd %.1f rev %.1f tb %.1f all %.1f fwdk %.1f | build %.1f chain %.1f gather %.1f | e2e %.0f reads/s"
          % (rep, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, plan.n_pairs / n, cells / n / 1e6,
             ms[0], ms[1], ms[2], ms[3], ms[4], mm[0], mm[1], mm[2], n / (t4 - t0)), flush=True)
    plan.close()
