"""Development aid: how much work the start pass and the traceback do on the bench workload, by class (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq
hwq.export_hw_queues()
import numpy as np
import torch  # noqa: F401
from ultra_amd import synth, lib, index as native
from ultra_amd import classify_read_with_mams as stage

world = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
w = synth.make_world(seed=20241008, **synth.WORLDS[world])
flat = synth.FlatWorld(w)
rs = synth.make_reads_native(flat, n, seed=1000, preset="ont_cdna", threads=8)
ix = native.NativeIndex(*w.index())
plan = ix.plan_arrays(rs.nam_off, rs.nam_part.reshape(-1, 3), rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
ctx = lib.Context(0)
stage.stage_batch(ctx, plan)
stage.run_staged(ctx, 0.6)
r = ctx.myers_fetch()
m = np.asarray(plan.packed["q_len"]).astype(np.int64); tl = np.asarray(plan.packed["t_len"]).astype(np.int64)
ed = np.asarray(r["ed"]).astype(np.int64); nl = np.asarray(r["n_loc"]).astype(np.int64)
hit = ed >= 0
fwd = (((m + 63) // 64) * 64 * tl).sum()
print("pairs %d hits %d locations %d  forward cells (64-row padded) %.3e" % (len(m), hit.sum(), nl[hit].sum(), fwd))
cols = m + ed
full = ((m + 63) // 64) * 64 * cols * nl
rows = np.select([2 * ed <= 22, 2 * ed <= 54, 2 * ed <= 118, 2 * ed <= 246, 2 * ed <= 374], [32, 64, 128, 256, 384], 0)
band = np.where(rows > 0, rows, ((m + 63) // 64) * 64) * cols * nl
print("start pass cells: full height %.3e (%.1f%% of forward)   banded %.3e (%.1f%%)" % (full[hit].sum(), 100 * full[hit].sum() / fwd, band[hit].sum(), 100 * band[hit].sum() / fwd))
for name, lo, hi in (("m<=64", 1, 64), ("65..128", 65, 128), ("129..256", 129, 256), ("257..512", 257, 512), ("513..2048", 513, 2048), (">2048", 2049, 10 ** 9)):
    s = hit & (m >= lo) & (m <= hi)
    if s.any():
        print("  %-10s pairs %8d hits %8d locs %9d  full %.3e  banded %.3e  fallback locs %d (cells %.3e)  mean ed/m %.3f" % (
            name, ((m >= lo) & (m <= hi)).sum(), s.sum(), nl[s].sum(), full[s].sum(), band[s].sum(), nl[s & (rows == 0)].sum(), full[s & (rows == 0)].sum(), (ed[s] / m[s]).mean()))
print("band class of locations:", {int(k): int(nl[hit & (rows == k)].sum()) for k in (32, 64, 128, 256, 384, 0)})
print("nloc > 32: pairs %d (of %d hits), by class m<=64: %d, 65..128: %d" % ((nl > 32).sum(), hit.sum(), ((nl > 32) & (m <= 64)).sum(), ((nl > 32) & (m > 64) & (m <= 128)).sum()))
# traceback (first location only): L x band(ed + 1 diagonals)
off = np.asarray(r["loc_off"]); ls, le = np.asarray(r["loc_start"]), np.asarray(r["loc_end"])
L = np.zeros(len(m), np.int64); first = off[:-1][hit]; L[hit] = le[first] - ls[first] + 1
trows = np.select([ed <= 22, ed <= 54, ed <= 118, ed <= 246, ed <= 374], [32, 64, 128, 256, 384], 0)
tb = np.where(trows > 0, trows, ((m + 63) // 64) * 64) * L
print("traceback cells %.3e (%.1f%% of forward); fallback hits %d with cells %.3e" % (tb[hit].sum(), 100 * tb[hit].sum() / fwd, (hit & (trows == 0)).sum(), tb[hit & (trows == 0)].sum()))
for k in (32, 64, 128, 256, 384, 0):
    s = hit & (trows == k)
    print("  tb rows %3d: hits %8d  cells %.3e  mean L %.0f" % (k, s.sum(), tb[s].sum(), L[s].mean() if s.any() else 0))
