import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq; hwq.export_hw_queues()
import numpy as np
from ultra_amd import synth, index, lib
from ultra_amd import classify_read_with_mams as stage
w = synth.make_world(seed=20241008, **dict(synth.WORLDS["c3"], n_genes=6000))
fw = synth.FlatWorld(w); rs = synth.make_reads_native(fw, 65536, seed=5, preset="ont_cdna")
ix = index.NativeIndex(*w.index()); ctx = lib.Context(0)
plan = ix.plan_arrays(rs.nam_off, rs.nam_part.reshape(-1, 3), rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
stage.stage_batch(ctx, plan); ctx.mam_run(0.6)
res = ctx.myers_fetch()
nl = res["n_loc"]; ed = res["ed"]; ql = plan.packed["q_len"]; tl = plan.packed["t_len"]; k = plan.packed["k"]
big = np.nonzero(nl > 32)[0]
print("overflowing pairs:", len(big))
for p in big[np.argsort(-ql[big])][:12]:
    q = plan.packed["arena"][plan.packed["q_off"][p]:plan.packed["q_off"][p] + ql[p]].tobytes()[:60]
    print("m", ql[p], "n", tl[p], "k", k[p], "ed", ed[p], "nloc", nl[p], q)
print("q_len of overflowing: ", np.sort(ql[big])[-20:])
