"""Development aid (CPU only): time of ultra_front_prepare (seed text -> instances + bounds, the part of the front end in
front of the device) on synthetic cDNA reads.   python scripts/dev/front_cpu.py [reads] [threads]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ultra_amd import synth, lib as _lib, front as front_mod
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = synth.make_world(seed=20241008, **synth.WORLDS["c3"])
flat = synth.FlatWorld(w)
rs = synth.make_reads_native(flat, n, seed=1000, preset="ont_cdna", threads=8)
text, line_off = synth.seed_text(rs)
none = np.zeros(len(rs), np.int64)
pb = front_mod.PackedBatch(rs.reads, rs.read_off, text, line_off[:-1], line_off[1:], none, none)
fr = front_mod.Front(None, None, None, {c: 200000 for c in range(synth.WORLDS["c3"]["n_chr"])}, n_threads=thr)
inp = fr._input(pb, 1200000, 1.0, 1, 8, 0.0)
v = _lib.FrontView()
for i in range(5):
    t0 = time.perf_counter()
    rc = fr.lib.ultra_front_prepare(fr._h, C.byref(inp), C.byref(v))
    dt = time.perf_counter() - t0
    print("rc", rc, "prepare %.1f ms for %d reads (%.0f bytes of text per read), %d threads: %.2f M reads/s; instances %d NAMs %d"
          % (dt * 1e3, n, text.size / n, thr, n / dt / 1e6, v.n_inst, v.n_nam))
