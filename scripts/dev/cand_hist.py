"""Development aid: candidates per instance on the bench workload (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq
hwq.export_hw_queues()
import numpy as np
import torch  # noqa: F401
from ultra_amd import synth, lib, index as native
w = synth.make_world(seed=20241008, **synth.WORLDS["c3"])
flat = synth.FlatWorld(w)
rs = synth.make_reads_native(flat, 131072, seed=1000, preset="ont_cdna", threads=8)
nix = native.NativeIndex(*w.index()); dix = native.DeviceIndex(nix, 0)
ctx = lib.Context(0)
ctx.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
p = ctx.mam_fetch_plan()
c = np.diff(p["inst_cand_off"])
print("instances", len(c), "candidates mean %.1f max %d" % (c.mean(), c.max()))
for q in (50, 90, 99, 99.9, 99.99):
    print("  p%s = %d" % (q, np.percentile(c, q)))
for t in (32, 64, 128, 256):
    print("  share of instances with <= %d candidates: %.4f (of candidates: %.4f)" % (t, (c <= t).mean(), c[c <= t].sum() / c.sum()))
