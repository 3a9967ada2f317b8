"""Development aid (GPU box): distribution of candidates / pairs / MAM slots per instance of a bench batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ultra_amd import hwq
hwq.export_hw_queues()
from ultra_amd import synth, lib, index as native_index
preset = sys.argv[1] if len(sys.argv) > 1 else "ont_cdna"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
w = synth.make_world(seed=20241008, **synth.WORLDS["c3"])
flat = synth.FlatWorld(w)
rs = synth.make_reads_native(flat, n, seed=1000, preset=preset, threads=16)
import torch
nix = native_index.NativeIndex(*w.index())
dix = native_index.DeviceIndex(nix, 0)
c = lib.Context(0)
c.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
p = c.mam_fetch_plan()
nc = np.diff(p["inst_cand_off"])
print("candidates per instance: mean %.1f max %d" % (nc.mean(), nc.max()), "percentiles 50/90/99/99.9:", np.percentile(nc, [50, 90, 99, 99.9]))
c.mam_run(synth.PRESETS[preset]["min_acc"])
r = c.mam_fetch()
nm = r["n_mam"]
print("MAMs per instance: mean %.1f max %d" % (nm.mean(), nm.max()), "percentiles 50/90/99/99.9:", np.percentile(nm, [50, 90, 99, 99.9]))
print("share <=12 / <=16 / <=24 / <=48:", [(nm <= t).mean().round(4) for t in (12, 16, 24, 48)])
cl = np.diff(r["chain_off"])
print("chain length: mean %.1f max %d" % (cl.mean(), cl.max()))
