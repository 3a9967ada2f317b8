"""Development aid: distribution of hits over the traceback classes for the bench workload (run on the GPU box)."""
import os
os.environ["GPU_MAX_HW_QUEUES"] = "13"  # fixed literal, within the pool limit
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from ultra_amd import synth, lib, index as native
from ultra_amd import classify_read_with_mams as stage

w = synth.make_world(seed=20241008, n_genes=1500, n_chr=8)
insts = synth.make_instances(w, 32768, seed=1000, preset="ont_cdna")
ix = native.NativeIndex(*w.index())
plan = ix.plan([(sol, read) for read, sol, _ in insts])
ctx = lib.Context(0)
stage.stage_batch(ctx, plan)
stage.run_staged(ctx, 0.6)
r = ctx.myers_fetch()
ed = np.asarray(r["ed"])
hit = ed >= 0
off = np.asarray(r["loc_off"])
L = np.zeros(len(ed), np.int64)
ls, le = np.asarray(r["loc_start"]), np.asarray(r["loc_end"])
first = off[:-1][hit]
L[hit] = le[first] - ls[first] + 1
print("pairs", len(ed), "hits", int(hit.sum()))
for lo, hi in [(0, 22), (23, 54), (55, 118), (119, 246), (247, 374), (375, 10**9)]:
    m = hit & (ed >= lo) & (ed <= hi)
    if m.any():
        print("ed %d..%d: hits %d  sum L %d  mean L %.0f  max L %d" % (lo, hi, m.sum(), L[m].sum(), L[m].mean(), L[m].max()))
res = ctx.mam_fetch()
nm = np.asarray(res["n_mam"])
print("instances", len(nm), "MAMs", int(nm.sum()), "mean", nm.mean(), "max", nm.max())
for b in (0, 4, 8, 16, 24, 32, 48, 64, 128, 256):
    print("  n_mam <= %d: %.4f   share of sum n^2 above: %.3f" % (b, (nm <= b).mean(), (nm[nm > b].astype(np.float64) ** 2).sum() / max(1.0, (nm.astype(np.float64) ** 2).sum())))
# hits whose insertion count follows from arithmetic: ed - |m - L| <= 1 leaves no choice (X = ed - |m - L|, I = max(m - L, 0))
m = np.asarray(plan.packed["q_len"]).astype(np.int64)
d = np.abs(m - L)
free = hit & (ed - d <= 1)
print("hits with ed - |m - L| <= 1: %d of %d (%.3f), their share of sum L %.3f, of sum L * (2 ed + 3) %.3f" % (
    free.sum(), hit.sum(), free.sum() / hit.sum(), L[free].sum() / L[hit].sum(),
    (L[free] * (2 * ed[free] + 3)).sum() / (L[hit] * (2 * ed[hit] + 3)).sum()))
for lo, hi in [(0, 0), (1, 1), (2, 2), (3, 5), (6, 22)]:
    mm = hit & (ed >= lo) & (ed <= hi)
    print("  ed %d..%d: hits %.3f of all, free among them %.3f" % (lo, hi, mm.sum() / hit.sum(), (free & mm).sum() / max(1, mm.sum())))
