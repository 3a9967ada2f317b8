"""Development aid (GPU box): a large batch against the same reads run as smaller batches - results of a read do not
depend on the batch it rides in.   python scripts/dev/big_batch.py 1048576 [parts]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ultra_amd import hwq  # noqa: E402

hwq.export_hw_queues()
from ultra_amd import index as native_index  # noqa: E402
from ultra_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
preset = sys.argv[3] if len(sys.argv) > 3 else "ont_cdna"
MYERS = len(sys.argv) > 4 and sys.argv[4] == "myers"     # also the Myers results pair by pair (first part only)
P = synth.PRESETS[preset]
w = synth.make_world(seed=20241008, **dict(synth.WORLDS["c3"]))
idx = w.index()
flat = synth.FlatWorld(w)
rs = synth.make_reads_native(flat, N, seed=1000, preset=preset, threads=16)
import torch  # noqa: E402,F401
from ultra_amd import lib  # noqa: E402

nix = native_index.NativeIndex(*idx)
dix = native_index.DeviceIndex(nix, 0)
FIELDS_C = ("x", "y", "c", "d", "msl", "cand", "j", "val")
FIELDS_M = ("x", "y", "c", "d", "msl", "cand", "val")


def run(lo, hi):
    c = lib.Context(0)
    n0, n1 = int(rs.nam_off[lo]), int(rs.nam_off[hi])
    r0, r1 = int(rs.read_off[lo]), int(rs.read_off[hi])
    part = np.asarray(rs.nam_part).reshape(-1, 3)[n0:n1]
    c.mam_stage_nams(dix, np.asarray(rs.nam_off[lo:hi + 1]) - n0, part, rs.nam_x[n0:n1], rs.nam_y[n0:n1],
                     rs.reads[r0:r1], np.asarray(rs.read_off[lo:hi + 1]) - r0)
    counts = c.mam_counts()
    c.mam_run(P["min_acc"])
    r = c.mam_fetch(want_all=True, want_names=True)
    cells = c.myers_cell_counts()
    if MYERS and (lo == 0):
        r["myers"] = c.myers_fetch()
    del c
    return counts, cells, r


cw, cellw, whole = run(0, N)
print("whole: counts", cw, "cells", cellw, "chains", int(whole["chain_off"][-1]), "mams", int(whole["mam_off"][-1]), flush=True)
step = N // parts
tot_c = tot_m = 0
bad = 0
for p in range(parts):
    lo, hi = p * step, (p + 1) * step
    cp, cellp, r = run(lo, hi)
    tot_c += int(r["chain_off"][-1]); tot_m += int(r["mam_off"][-1])
    co, mo = int(whole["chain_off"][lo]), int(whole["mam_off"][lo])
    msg = []
    if not np.array_equal(whole["n_mam"][lo:hi], r["n_mam"]):
        d = np.flatnonzero(whole["n_mam"][lo:hi] != r["n_mam"])
        msg.append("n_mam differs at %d reads, first %d (whole %d part %d)" % (len(d), lo + d[0], whole["n_mam"][lo + d[0]], r["n_mam"][d[0]]))
    if whole["value"][lo:hi].tobytes() != r["value"].tobytes():
        d = np.flatnonzero(whole["value"][lo:hi] != r["value"])
        msg.append("value differs at %d reads, first %d (whole %r part %r)" % (len(d), lo + d[0], whole["value"][lo + d[0]], r["value"][d[0]]))
    if not np.array_equal(whole["chain_off"][lo:hi + 1] - co, r["chain_off"]):
        msg.append("chain_off differs")
    else:
        n = int(r["chain_off"][-1])
        for f in FIELDS_C:
            if f == "cand" and lo:      # candidate numbers count through the batch
                continue
            if whole["chain_" + f][co:co + n].tobytes() != r["chain_" + f][:n].tobytes():
                d = np.flatnonzero(whole["chain_" + f][co:co + n] != r["chain_" + f][:n])
                msg.append("chain_%s differs at %d elements, first %d" % (f, len(d), d[0]))
    if not np.array_equal(whole["mam_off"][lo:hi + 1] - mo, r["mam_off"]):
        msg.append("mam_off differs")
    else:
        n = int(r["mam_off"][-1])
        for f in FIELDS_M:
            if f == "cand" and lo:
                continue
            if whole["mam_" + f][mo:mo + n].tobytes() != r["mam_" + f][:n].tobytes():
                d = np.flatnonzero(whole["mam_" + f][mo:mo + n] != r["mam_" + f][:n])
                msg.append("mam_%s differs at %d elements, first %d" % (f, len(d), d[0]))
    if MYERS and p == 0:
        a, b = whole["myers"], r["myers"]
        n = len(b["ed"])
        print("myers: pairs of part 0", n, " hits whole", int((a["ed"][:n] >= 0).sum()), "part", int((b["ed"] >= 0).sum()),
              " hits of the whole batch", int((a["ed"] >= 0).sum()), "locations", int(a["loc_off"][-1]))
        for f in ("ed", "n_loc", "first_ins"):
            d = np.flatnonzero(a[f][:n] != b[f])
            print("  %s differs at %d pairs%s" % (f, len(d), "" if not len(d) else ", first %d (whole %d part %d), part's ed there: min %d max %d"
                                                % (d[0], a[f][d[0]], b[f][d[0]], b["ed"][d].min(), b["ed"][d].max())))
        if np.array_equal(a["loc_off"][:n + 1], b["loc_off"]):
            m = int(b["loc_off"][-1])
            for f in ("loc_start", "loc_end"):
                d = np.flatnonzero(a[f][:m] != b[f][:m])
                print("  %s differs at %d locations" % (f, len(d)))
        else:
            print("  loc_off differs")
    print("part %d [%d, %d): counts %s cells %s  %s" % (p, lo, hi, cp, cellp, "; ".join(msg) if msg else "equal"), flush=True)
    bad += bool(msg)
print("parts: chains", tot_c, "mams", tot_m)
print("RESULT", "DIFFERENT" if bad else "EQUAL")
