"""GPU: candidate enumeration on the device (csrc/dplan.hip, ultra_mam_stage_nams) against the host planner
(csrc/plan.hip, itself pinned against the Python port of the reference's dictionary walk in tests/test_plan_native.py):
the staged batches must agree candidate by candidate and pair by pair (by content - the arenas differ: the host planner
copies the sequences of a batch, the device planner points into the resident pool), and the results of the whole MAM
stage must be the same tuples, equal to the oracle's."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (a[0] == b[0] and struct.pack("d", float(a[1])) == struct.pack("d", float(b[1]))
            and tuple(tuple(m) for m in a[2]) == tuple(tuple(m) for m in b[2]))


def _classes(ids, members):
    """Partition of `members` by id value, as a set of frozensets."""
    by = {}
    for i in members:
        by.setdefault(int(ids[i]), []).append(i)
    return {frozenset(v) for v in by.values()}


def _compare_plans(host, host_arena, dev, dev_arena):
    for f in ("inst_cand_off", "inst_read_len", "inst_first_part_stop", "inst_last_part_start", "cand_kind", "cand_e_start",
              "cand_e_stop", "cand_seq_len", "cand_chr"):
        assert np.array_equal(host[f], dev[f]), f
    assert len(host["k"]) == len(dev["k"])          # the same number of distinct pairs
    def content(p, arena, i):
        if i < 0:
            return None
        return (arena[p["q_off"][i]:p["q_off"][i] + p["q_len"][i]].tobytes(), arena[p["t_off"][i]:p["t_off"][i] + p["t_len"][i]].tobytes(),
                int(p["k"][i]))
    for c in range(len(host["cand_kind"])):
        for f in ("cand_pair", "cand_pair_contained"):
            assert content(host, host_arena, int(host[f][c])) == content(dev, dev_arena, int(dev[f][c])), (c, f)
    # sequence identity classes matter inside the partial-segment loop and inside the partial-flank loop (:479-523)
    off = host["inst_cand_off"]
    for i in range(len(off) - 1):
        rng = range(int(off[i]), int(off[i + 1]))
        for kinds in ((2, 3), (4, 5)):
            members = [c for c in rng if host["cand_kind"][c] in kinds]
            assert _classes(host["cand_seq_id"], members) == _classes(dev["cand_seq_id"], members), (i, kinds)


@pytest.mark.parametrize("preset,kw", [("ont_cdna", {}), ("ont_drna", {"small_exon_frac": 0.3, "split_frac": 0.7}),
                                       ("isoseq", {"overlap_frac": 0.4, "exon_mean": 12.0, "max_exons": 60})])
def test_device_plan_equals_host_plan(ctx, preset, kw):
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import synth
    w = synth.make_world(seed=61, n_genes=60, n_chr=2, **kw)
    idx = w.index()
    insts = synth.make_instances(w, 240, seed=17, preset=preset, flank_start_prob=0.35, flank_end_prob=0.35)
    pairs = [(sol, read) for read, sol, _ in insts] + [([], insts[0][0])]      # + an instance without NAMs
    min_acc = synth.PRESETS[preset]["min_acc"]
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    try:
        flat = native.flatten_instances(pairs)
        plan = ix.plan_arrays(*flat)
        stage.stage_batch(ctx, plan)
        host = ctx.mam_fetch_plan()
        host_arena = plan.packed["arena"].copy()
        stage.run_staged(ctx, min_acc)
        want = stage.results_to_python(plan, ctx.mam_fetch())
        plan.close()
        ctx.mam_stage_nams(dix, *flat)
        dev = ctx.mam_fetch_plan()
        dev_arena = np.concatenate([ix.pool[:ix.pool_bytes], flat[4]])
        _compare_plans(host, host_arena, dev, dev_arena)
        stage.run_staged(ctx, min_acc)
        got = stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist())
        # a second batch on the same context reuses the resident pool
        ctx.mam_stage_nams(dix, *native.flatten_instances(pairs[:50]))
        stage.run_staged(ctx, min_acc)
        got2 = stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist()[:50])
    finally:
        dix.close(); ix.close()
    assert len(got) == len(want) == len(pairs)
    for g, w_ in zip(got, want):
        assert _same(g, w_)
    for g, w_ in zip(got2, want[:50]):
        assert _same(g, w_)
    for i in range(0, len(insts), 7):
        read, sol, _ = insts[i]
        assert _same(classify_port.classify(sol, *idx, read, None, min_acc), got[i])
    assert got[-1] == ([], -1, ())


def test_device_plan_big_instances_and_errors(ctx):
    """A gene of 700 small segments overflows the small-capacity kernel (re-run at the large capacity); a NAM on an
    unknown part is the reference's KeyError."""
    from array import array
    from oracle import classify_port
    from tests import helpers
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import lib
    from ultra_amd.synth import Mem, key_of
    rng = np.random.RandomState(8)
    seg, flank, p2s, s2g, g2s = {}, {}, {}, {}, {"g": array('L'), "h": array('L')}
    pos, segs = 1000, []
    for i in range(700):
        ln = int(rng.choice([9, 10, 12, 15, 20, 30]))
        s = helpers.rand_seq(rng, ln, "ACGT")
        k = key_of(1, pos, pos + ln)
        g = "g" if i < 690 else "h"
        seg[k] = s; s2g[k] = {g}; p2s[k] = array('L', [1, pos, pos + ln]); g2s[g].extend([1, pos, pos + ln])
        segs.append((pos, pos + ln, s))
        pos += ln + 50
    flank[key_of(1, 0, 900)] = helpers.rand_seq(rng, 900, "ACGT")
    idx = (seg, flank, p2s, s2g, g2s)
    insts = []
    for r in range(6):
        picks = sorted(rng.choice(len(segs), size=12, replace=False))
        read = "".join(segs[i][2] for i in picks) + helpers.rand_seq(rng, 40, "ACGT")
        a, b, _ = segs[picks[0]]
        sol = [Mem(a, b - 1, 0, b - a - 1, b - a, 0, "1^%d^%d" % (a, b))]
        if r % 2:
            sol = [Mem(850, 899, 0, 49, 50, 0, "1^0^900")] + sol      # + a flank hit, partially covered
        insts.append((read, sol))
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    try:
        ctx.mam_stage_nams(dix, *native.flatten_instances([(sol, read) for read, sol in insts]))
        stage.run_staged(ctx, 0.6)
        got = stage.results_from_names(ctx.mam_fetch(want_names=True), [len(r) for r, _ in insts])
        for (read, sol), g in zip(insts, got):
            assert _same(classify_port.classify(sol, *idx, read, None, 0.6), g)
        bad = [([Mem(5, 20, 0, 15, 16, 0, "1^5^77")], insts[0][0])]
        with pytest.raises(lib.UltraError):
            ctx.mam_stage_nams(dix, *native.flatten_instances(bad))
        # the context still works afterwards
        ctx.mam_stage_nams(dix, *native.flatten_instances([(sol, read) for read, sol in insts[:2]]))
        stage.run_staged(ctx, 0.6)
        again = stage.results_from_names(ctx.mam_fetch(want_names=True), [len(r) for r, _ in insts[:2]])
        assert all(_same(a, b) for a, b in zip(again, got[:2]))
    finally:
        dix.close(); ix.close()


def test_pipelined_batches_equal_plain_runs(ctx):
    """ultra_mam_run_begin / ultra_mam_plan_resident_behind / ultra_mam_run_end: three batches on two contexts, the
    planner of batch i + 1 enqueued behind the start pass of batch i's run, give the results of plain plan + run
    (and of the oracle); state errors are reported, and an empty batch neither blocks nor breaks the chain."""
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import lib, synth
    w = synth.make_world(seed=77, n_genes=80, n_chr=2)
    idx = w.index()
    batches = []
    for b, n in enumerate((300, 0, 170, 260)):
        insts = synth.make_instances(w, n, seed=40 + b, preset="ont_cdna", flank_start_prob=0.3, flank_end_prob=0.3) if n else []
        batches.append((insts, native.flatten_instances([(sol, read) for read, sol, _ in insts])))
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    other = lib.Context(0)
    cs = [ctx, other]
    try:
        want = []
        for insts, flat in batches:
            ctx.mam_stage_nams(dix, *flat)
            ctx.mam_run(0.6)
            want.append(stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist()))
        with pytest.raises(lib.UltraError):
            ctx.mam_run_end()                       # nothing in flight
        got = [None] * len(batches)
        cs[0].mam_stage_nams(dix, *batches[0][1], upload_only=True)
        cs[0].mam_plan_resident(dix)
        for i, (insts, flat) in enumerate(batches):
            c, nxt = cs[i % 2], cs[(i + 1) % 2]
            c.mam_run_begin(0.6)
            with pytest.raises(lib.UltraError):
                c.mam_run_begin(0.6)                # the run in flight has not been closed
            if i + 1 < len(batches):
                nxt.mam_stage_nams(dix, *batches[i + 1][1], upload_only=True)
                nxt.mam_plan_resident_behind(dix, c)
            c.mam_run_end()
            got[i] = stage.results_from_names(c.mam_fetch(want_names=True), np.diff(flat[5]).tolist())
        # behind a context with no run in flight it is the plain planner
        cs[1].mam_stage_nams(dix, *batches[0][1], upload_only=True)
        cs[1].mam_plan_resident_behind(dix, cs[0])
        cs[1].mam_run(0.6)
        again = stage.results_from_names(cs[1].mam_fetch(want_names=True), np.diff(batches[0][1][5]).tolist())
    finally:
        other.close(); dix.close(); ix.close()
    for g, w_ in zip(got, want):
        assert len(g) == len(w_)
        for a, b in zip(g, w_):
            assert _same(a, b)
    for a, b in zip(again, want[0]):
        assert _same(a, b)
    insts = batches[0][0]
    for i in range(0, len(insts), 9):
        read, sol, _ = insts[i]
        assert _same(classify_port.classify(sol, *idx, read, None, 0.6), got[0][i])


@pytest.mark.parametrize("preset", ["ont_cdna", "ont_drna"])
def test_results_do_not_depend_on_the_batch(preset):
    """Size-independent property at a bench-like size: 98 304 reads as one batch give, read by read, the values, chains and
    MAM lists of the same reads as three batches (scripts/dev/big_batch.py does the same with a million: that is how the
    24-bit hit offsets of the first three rounds were found)."""
    from ultra_amd import index as native
    from ultra_amd import lib, synth
    w = synth.make_world(seed=20241008, **dict(synth.WORLDS["small"]))
    rs = synth.make_reads_native(synth.FlatWorld(w), 98304, seed=7, preset=preset, threads=8)
    ix = native.NativeIndex(*w.index())
    dix = native.DeviceIndex(ix)
    min_acc = synth.PRESETS[preset]["min_acc"]

    def run(lo, hi):
        c = lib.Context(0)
        try:
            n0, n1 = int(rs.nam_off[lo]), int(rs.nam_off[hi])
            r0, r1 = int(rs.read_off[lo]), int(rs.read_off[hi])
            c.mam_stage_nams(dix, np.asarray(rs.nam_off[lo:hi + 1]) - n0, np.asarray(rs.nam_part).reshape(-1, 3)[n0:n1],
                             rs.nam_x[n0:n1], rs.nam_y[n0:n1], rs.reads[r0:r1], np.asarray(rs.read_off[lo:hi + 1]) - r0)
            c.mam_run(min_acc)
            return c.mam_fetch(want_all=True, want_names=True)
        finally:
            c.close()
    try:
        whole = run(0, len(rs))
        assert int(whole["mam_off"][-1]) > 5 * len(rs)
        step = len(rs) // 3
        for p in range(3):
            lo, hi = p * step, (p + 1) * step
            part = run(lo, hi)
            assert np.array_equal(whole["n_mam"][lo:hi], part["n_mam"])
            assert whole["value"][lo:hi].tobytes() == part["value"].tobytes()
            for kind, fields in (("chain", ("x", "y", "c", "d", "msl", "j", "val", "kind", "chr")), ("mam", ("x", "y", "c", "d", "msl", "val"))):
                o = int(whole[kind + "_off"][lo])
                assert np.array_equal(whole[kind + "_off"][lo:hi + 1] - o, part[kind + "_off"])
                n = int(part[kind + "_off"][-1])
                for f in fields:
                    assert whole["%s_%s" % (kind, f)][o:o + n].tobytes() == part["%s_%s" % (kind, f)][:n].tobytes(), (kind, f, p)
    finally:
        dix.close(); ix.close()
