"""Development aid: where a batch's wall time goes on a C3-shaped workload: host plan + stage vs device plan."""
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
t = time.time(); dix = index.DeviceIndex(ix); print("device index %.2fs" % (time.time() - t), flush=True)
ctx = lib.Context(0)
min_acc = synth.PRESETS[preset]["min_acc"]
ref = None
for rep in range(1):
    t0 = time.time(); plan = ix.plan_arrays(rs.nam_off, rs.nam_part.reshape(-1, 3), rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
    t1 = time.time(); stage.stage_batch(ctx, plan)
    t2 = time.time(); stage.run_staged(ctx, min_acc)
    t3 = time.time(); res = ctx.mam_fetch(want_names=True)
    t4 = time.time()
    cells, ms = ctx.myers_timing(); mm = ctx.mam_timing()
    print("host rep %d: plan %.0f ms (%.1f us/read)  stage %.0f ms  run %.1f ms  fetch %.1f ms | pairs/read %.1f cells/read %.2fM | myers fwd %.1f rev %.1f tb %.1f all %.1f fwdk %.1f | build %.1f chain %.1f gather %.1f | e2e %.0f reads/s"
          % (rep, (t1 - t0) * 1e3, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, plan.n_pairs / n, cells / n / 1e6,
             ms[0], ms[1], ms[2], ms[3], ms[4], mm[0], mm[1], mm[2], n / (t4 - t0)), flush=True)
    plan.close()
    ref = res
for rep in range(4):
    t1 = time.time(); ctx.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
    t2 = time.time(); stage.run_staged(ctx, min_acc)
    t3 = time.time(); res = ctx.mam_fetch(want_names=True)
    t4 = time.time()
    cells, ms = ctx.myers_timing(); mm = ctx.mam_timing()
    print("dev rep %d: plan+stage %.1f ms  run %.1f ms  fetch %.1f ms | cells/read %.2fM | myers fwd %.1f rev %.1f tb %.1f all %.1f | e2e %.0f reads/s"
          % (rep, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, cells / n / 1e6, ms[0], ms[1], ms[2], ms[3], n / (t4 - t1)), flush=True)
same = all(np.array_equal(ref[k], res[k]) for k in ref if k != "chain_cand")
print("device-planned results equal host-planned:", same)
# ---- pipeline: 4 sub-batches x 2 rounds through ultra_pipe ----
for depth in (1, 2, 3):
    pipe = lib.Pipe(ix, 0, depth=depth)
    nb = 16
    sub = [rs.slice(i * n // 4, (i + 1) * n // 4) for i in range(4)]
    def go(b):
        s = sub[b % 4]
        pipe.submit(s.nam_off, s.nam_part, s.nam_x, s.nam_y, s.reads, s.read_off, min_acc)
    # warm-up: every slot allocates its buffers on its first batches
    for b in range(2 * depth):
        go(b)
        if pipe.pending() >= depth:
            pipe.wait(); pipe.release()
    while pipe.pending():
        pipe.wait(); pipe.release()
    t0 = time.time()
    sent = done = 0
    tims = []
    while done < nb:
        while sent < nb and sent - done < depth:
            go(sent); sent += 1
        r = pipe.wait(); tims.append(pipe.timing()); pipe.release(); done += 1
    dt = time.time() - t0
    avg = lambda k: sum(t[k] for t in tims) / len(tims)
    print("pipe depth %d: %d batches of %d reads in %.1f ms -> %.0f reads/s | per batch: stage %.1f wait %.1f run %.1f fetch %.1f ms"
          % (depth, nb, n // 4, dt * 1e3, nb * (n // 4) / dt, avg("stage_ms"), avg("wait_ms"), avg("run_ms"), avg("fetch_ms")), flush=True)
    pipe.close()
