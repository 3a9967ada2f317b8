"""Development aid (GPU box): device memory one context holds after a batch of N reads.   python scripts/dev/mem_probe.py 1048576 [preset]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq  # noqa: E402

hwq.export_hw_queues()
from ultra_amd import index as native_index  # noqa: E402
from ultra_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
preset = sys.argv[2] if len(sys.argv) > 2 else "ont_cdna"
w = synth.make_world(seed=20241008, **dict(synth.WORLDS["c3"]))
rs = synth.make_reads_native(synth.FlatWorld(w), N, seed=1000, preset=preset, threads=16)
import torch  # noqa: E402
from ultra_amd import lib  # noqa: E402

used = lambda: (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30
torch.cuda.init()
u0 = used()
dix = native_index.DeviceIndex(native_index.NativeIndex(*w.index()), 0)
u1 = used()
c = lib.Context(0)
c.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
u2 = used()
c.mam_run(synth.PRESETS[preset]["min_acc"])
u3 = used()
print("GiB: before %.2f  index +%.2f  staged +%.2f  after the run +%.2f  (context total %.2f)" % (u0, u1 - u0, u2 - u1, u3 - u2, u3 - u1))
