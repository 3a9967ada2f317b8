"""GPU parity of the whole MAM stage: ultra_amd.classify_read_with_mams (fused HIP pipeline through
the C ABI: Myers -> MAM lists -> chaining) vs (a) the golden vectors produced by the reference's own
module and (b) the CPU oracle on larger seeded batches."""
import json
import os
import struct

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "classify_golden.json")


def _same(a, b):
    return (a[0] == b[0] and struct.pack("d", float(a[1])) == struct.pack("d", float(b[1]))
            and tuple(tuple(m) for m in a[2]) == tuple(tuple(m) for m in b[2]))


def test_main_batch_matches_reference_golden_vectors(ctx):
    from ultra_amd import classify_read_with_mams as stage
    gold = json.load(open(GOLD))
    n = 0
    for w in gold["worlds"]:
        idx = helpers.load_golden_world(w["world"])
        by_acc = {}
        for inst in w["instances"]:
            by_acc.setdefault(inst["min_acc"], []).append(inst)
        for min_acc, insts in by_acc.items():
            got = stage.main_batch([(helpers.golden_solution(i["solution"]), i["read"]) for i in insts], *idx,
                                   min_acc=min_acc, ctx=ctx)
            for inst, (gaps, value, mams) in zip(insts, got):
                assert [int(g) for g in gaps] == inst["non_covered_regions"]
                assert float(value).hex() == inst["value"]
                assert helpers.mam_rows(mams) == inst["mam_solution"]
                n += 1
    assert n >= 140


def test_add_segment_to_mam_matches_reference_golden_vectors(ctx):
    from ultra_amd import classify_read_with_mams as stage
    gold = json.load(open(GOLD))
    for c in gold["add_segment_to_mam"]:
        out = []
        stage.add_segment_to_mam(c["read"], 3, c["seq"], c["e_start"], c["e_stop"], c["segm_id"], out, c["min_acc"], c["label"], ctx=ctx)
        assert helpers.mam_rows(out) == c["mams"]


@pytest.mark.parametrize("preset,n", [("ont_cdna", 300), ("isoseq", 120), ("ont_drna", 200)])
def test_main_batch_matches_oracle(ctx, preset, n):
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import synth
    w = synth.make_world(seed=21, n_genes=80)
    insts = synth.make_instances(w, n, seed=4, preset=preset)
    idx = w.index()
    min_acc = synth.PRESETS[preset]["min_acc"]
    got = stage.main_batch([(sol, read) for read, sol, _ in insts], *idx, min_acc=min_acc, ctx=ctx)
    bad = 0
    for (read, sol, _), g in zip(insts, got):
        want = classify_port.classify(sol, *idx, read, None, min_acc)
        if not _same(want, g):
            bad += 1
    assert bad == 0


@pytest.mark.parametrize("kw", [{}, {"small_exon_frac": 0.4, "split_frac": 0.7}])
def test_full_mam_lists_match_oracle(ctx, kw):
    """MAM SETS, not only the chosen chains: every instance's filtered, sorted MAM list."""
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import synth
    w = synth.make_world(seed=23, n_genes=40, **kw)
    insts = synth.make_instances(w, 150, seed=6, preset="ont_drna", flank_start_prob=0.4, flank_end_prob=0.4)
    idx = w.index()
    batch = stage.prepare_batch([(sol, read) for read, sol, _ in insts], *idx)
    stage.stage_batch(ctx, batch)
    stage.run_staged(ctx, 0.6)
    res = ctx.mam_fetch(want_all=True)
    lists = stage.mam_lists_to_python(batch, res)
    n_mams = 0
    for (read, sol, _), got in zip(insts, lists):
        want = classify_port.mam_instance(sol, *idx, read, 0.6)
        assert helpers.mam_rows(got) == helpers.mam_rows(want)
        n_mams += len(want)
    assert n_mams > 1000
    # a second run over the same staged batch gives the same bytes (no state leaks between runs)
    stage.run_staged(ctx, 0.6)
    res2 = ctx.mam_fetch(want_all=True)
    for k in res:
        assert np.array_equal(res[k], res2[k]), k


def test_low_complexity_reads_many_locations(ctx):
    """Repeats give long location lists, repeated starts and 5-8 bp exact segments with many hits."""
    from array import array
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd.synth import Mem, key_of
    rng = np.random.RandomState(3)
    insts, seg, flank, p2s, s2g, g2s = [], {}, {}, {}, {}, {"g": array('L')}
    pos = 1000
    segs = []
    for i in range(24):
        ln = int(rng.choice([5, 6, 7, 8, 9, 12, 20, 40, 90]))
        s = helpers.rand_seq(rng, ln, "AC" if i % 2 else "A")
        k = key_of(1, pos, pos + ln)
        seg[k] = s; s2g[k] = {"g"}; p2s[k] = array('L', [1, pos, pos + ln]); g2s["g"].extend([1, pos, pos + ln])
        segs.append((pos, pos + ln, s))
        pos += ln + 300
    for r in range(40):
        read = helpers.rand_seq(rng, int(rng.randint(150, 500)), "AC" if r % 3 else "A")
        a, b, s = segs[int(rng.randint(0, len(segs)))]
        sol = [Mem(a, b - 1, 0, b - a - 1, b - a, 0, "1^%d^%d" % (a, b))]
        insts.append((read, sol))
    idx = (seg, flank, p2s, s2g, g2s)
    got = stage.main_batch([(sol, read) for read, sol in insts], *idx, min_acc=0.6, ctx=ctx)
    for (read, sol), g in zip(insts, got):
        assert _same(classify_port.classify(sol, *idx, read, None, 0.6), g)


def test_solver_switch_is_taken_on_the_device(monkeypatch):
    """The chaining launches are made up from slot bounds on the host and the actual MAM counts on the device
    (classify_read_with_mams.py:560-575: more than 200 MAMs -> n log n solver).  With the switch lowered to 6 on both
    sides, instances land on either side of it with bounds above it; a threshold the n log n solver cannot take makes
    the library read the exact counts back and fail exactly when such an instance exists."""
    from array import array
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage, lib
    from ultra_amd.synth import Mem, key_of
    rng = np.random.RandomState(11)
    insts, seg, flank, p2s, s2g, g2s = [], {}, {}, {}, {}, {"g": array('L')}
    pos, segs = 1000, []
    for i in range(30):
        ln = int(rng.choice([6, 8, 12, 20, 40, 90]))
        s = helpers.rand_seq(rng, ln, "AC" if i % 2 else "ACGT")
        k = key_of(1, pos, pos + ln)
        seg[k] = s; s2g[k] = {"g"}; p2s[k] = array('L', [1, pos, pos + ln]); g2s["g"].extend([1, pos, pos + ln])
        segs.append((pos, pos + ln, s))
        pos += ln + 300
    for r in range(60):
        read = helpers.rand_seq(rng, int(rng.randint(60, 400)), "AC" if r % 3 else "ACGT")
        a, b, s = segs[int(rng.randint(0, len(segs)))]
        insts.append((read, [Mem(a, b - 1, 0, b - a - 1, b - a, 0, "1^%d^%d" % (a, b))]))
    idx = (seg, flank, p2s, s2g, g2s)
    monkeypatch.setattr(classify_port, "SWITCH_N", 6)
    want = [classify_port.classify(sol, *idx, read, None, 0.6) for read, sol in insts]
    sizes = [len(w[2]) for w in want]
    assert min(sizes) <= 6 < max(sizes)          # both solvers are taken
    c = lib.Context(0, cfg={"chain_switch_n": 6})
    try:
        got = stage.main_batch([(sol, read) for read, sol in insts], *idx, min_acc=0.6, ctx=c)
    finally:
        c.close()
    for w, g in zip(want, got):
        assert _same(w, g)
    c = lib.Context(0, cfg={"chain_switch_n": 6, "overlap_large": 0})
    try:
        with pytest.raises(lib.UltraError, match="overlap threshold"):
            stage.main_batch([(sol, read) for read, sol in insts], *idx, min_acc=0.6, ctx=c)
    finally:
        c.close()


def test_many_candidates_per_instance(ctx):
    """A gene of 1 300 short segments: every instance has more candidates than the two-instances-per-wavefront form
    of the MAM kernel keeps in LDS, so the batch takes the wavefront-per-instance form (several candidates per lane)."""
    from array import array
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd.synth import Mem, key_of
    rng = np.random.RandomState(8)
    seg, flank, p2s, s2g, g2s = {}, {}, {}, {}, {"g": array('L')}
    pos, segs = 1000, []
    for i in range(1300):
        ln = int(rng.choice([9, 10, 12, 15, 20, 30]))
        s = helpers.rand_seq(rng, ln, "ACGT")
        k = key_of(1, pos, pos + ln)
        seg[k] = s; s2g[k] = {"g"}; p2s[k] = array('L', [1, pos, pos + ln]); g2s["g"].extend([1, pos, pos + ln])
        segs.append((pos, pos + ln, s))
        pos += ln + 50
    insts = []
    for r in range(6):
        picks = sorted(rng.choice(len(segs), size=12, replace=False))
        read = "".join(segs[i][2] for i in picks) + helpers.rand_seq(rng, 40, "ACGT")
        a, b, _ = segs[picks[0]]
        insts.append((read, [Mem(a, b - 1, 0, b - a - 1, b - a, 0, "1^%d^%d" % (a, b))]))
    idx = (seg, flank, p2s, s2g, g2s)
    got = stage.main_batch([(sol, read) for read, sol in insts], *idx, min_acc=0.6, ctx=ctx)
    n_mams = 0
    for (read, sol), g in zip(insts, got):
        assert _same(classify_port.classify(sol, *idx, read, None, 0.6), g)
        n_mams += len(g[2])
    assert n_mams > 0


def test_single_instance_signatures(ctx):
    from oracle import classify_port, cport
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import colinear_solver, edlib_shim, synth
    w = synth.make_world(seed=22, n_genes=20)
    read, sol, _ = synth.make_instances(w, 1, seed=5)[0]
    idx = w.index()
    got = stage.main(sol, *idx, read, None, 0.6)
    assert _same(classify_port.classify(sol, *idx, read, None, 0.6), got)
    assert stage.main_batch([], *idx, min_acc=0.6, ctx=ctx) == []
    # edlib-shaped call
    r = edlib_shim.align("ACGTTGCA", "TTACGTAGCATT", mode="HW", task="path", k=3)
    o = cport.hw_align("ACGTTGCA", "TTACGTAGCATT", 3)
    assert r["editDistance"] == o["editDistance"] and r["locations"] == o["locations"] and r["cigar"] == o["cigar"]
    assert edlib_shim.align("AAAAAAAA", "CCCCCCCCCC", mode="HW", task="path", k=2)["editDistance"] == -1
    r = edlib_shim.align("ACTG", "CACTRT", mode="HW", task="path")   # edlib's published README example
    assert (r["editDistance"], r["locations"], r["cigar"]) == (1, [(1, 3), (1, 4)], "3=1I")
    # chainers with the reference's signatures
    mams = list(got[2])
    if mams:
        sol_q, val_q, _ = colinear_solver.read_coverage_mam_score(mams, overlap_threshold=20)
        assert len(sol_q) >= 1


def test_bad_candidate_lists_fail_loudly(ctx):
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import lib, synth
    w = synth.make_world(seed=22, n_genes=10)
    insts = synth.make_instances(w, 4, seed=5)
    batch = stage.prepare_batch([(sol, read) for read, sol, _ in insts], *w.index())
    bad = dict(batch.packed)
    bad["cand_pair"] = batch.packed["cand_pair"].copy(); bad["cand_pair"][0] = batch.n_pairs + 5
    with pytest.raises(lib.UltraError):
        ctx.mam_stage(bad)
    bad = dict(batch.packed)
    bad["cand_kind"] = batch.packed["cand_kind"].copy(); bad["cand_kind"][0] = 3
    with pytest.raises(lib.UltraError):
        ctx.mam_stage(bad)


@pytest.mark.parametrize("preset", ["ont_cdna", "ont_drna"])
def test_native_plan_gives_the_same_results(ctx, preset):
    """Row f4: the batch planned inside the library (packed index) goes through the same device stage and returns
    the same tuples - names, chromosome ids, values, gaps - as the batch planned by the Python port."""
    from ultra_amd import classify_read_with_mams as hip_stage
    from ultra_amd import index as native
    from ultra_amd import synth
    w = synth.make_world(seed=41, n_genes=50)
    idx = w.index()
    insts = synth.make_instances(w, 200, seed=12, preset=preset, flank_start_prob=0.3, flank_end_prob=0.3)
    pairs = [(sol, read) for read, sol, _ in insts]
    min_acc = synth.PRESETS[preset]["min_acc"]
    want = hip_stage.main_batch(pairs, *idx, min_acc=min_acc, ctx=ctx)
    ix = native.NativeIndex(*idx)
    try:
        got = hip_stage.main_batch(pairs, *idx, min_acc=min_acc, ctx=ctx, index=ix)
    finally:
        ix.close()
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert g[0] == w_[0]
        assert struct.pack("d", float(g[1])) == struct.pack("d", float(w_[1]))
        assert tuple(tuple(m) for m in g[2]) == tuple(tuple(m) for m in w_[2])


def test_launch_lanes_and_empty_batches(ctx):
    """Results do not depend on how many streams the launches of a pass are dealt over (1, 4 or 8 lanes), and batches
    whose candidates yield no Myers pair at all (every segment narrower than 5 bp) or no candidate at all come back as
    the reference's empty result ([], -1, ()) - the slot layout that rides inside the Myers run must cope with both."""
    from array import array
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import lib, synth
    from ultra_amd.synth import Mem, key_of
    w = synth.make_world(seed=23, n_genes=30)
    idx = w.index()
    insts = synth.make_instances(w, 120, seed=7, preset="ont_drna", flank_start_prob=0.3, flank_end_prob=0.3)
    pairs = [(sol, read) for read, sol, _ in insts]
    want = stage.main_batch(pairs, *idx, min_acc=0.6, ctx=ctx)
    for lanes in (1, 4, 8):
        c = lib.Context(0, cfg={"launch_lanes": lanes})
        try:
            got = stage.main_batch(pairs, *idx, min_acc=0.6, ctx=c)
        finally:
            c.close()
        assert all(_same(a, b) for a, b in zip(want, got)), lanes
    # a world whose only segments are 3-4 bp wide: candidates exist, pairs do not
    seg = {key_of(1, 100, 104): "ACGT", key_of(1, 300, 303): "TTG"}
    s2g = {k: {"g"} for k in seg}
    p2s = {key_of(1, 90, 320): array('L', [1, 100, 104, 1, 300, 303])}
    g2s = {"g": array('L', [1, 100, 104, 1, 300, 303])}
    tiny = (seg, {}, p2s, s2g, g2s)
    read = "ACGTTTGACGTACGTACGATCGATCGTAGCTAGCTAGCATCGATCGAT"
    sols = [[Mem(100, 103, 0, 3, 4, 0, "1^90^320")], [Mem(300, 302, 10, 12, 3, 0, "1^90^320")], []]
    got = stage.main_batch([(s, read) for s in sols], *tiny, min_acc=0.6, ctx=ctx)
    for s, g in zip(sols, got):
        assert _same(classify_port.classify(s, *tiny, read, None, 0.6), g)
        assert g == ([], -1, ())
    # mixed: the same empty instances between real ones of another world cannot be batched (one index per batch),
    # but an instance with an empty NAM chain inside a real batch can
    mixed = pairs[:5] + [([], insts[0][0])] + pairs[5:10]
    got = stage.main_batch(mixed, *idx, min_acc=0.6, ctx=ctx)
    assert got[5] == ([], -1, ())
    assert all(_same(a, b) for a, b in zip(want[:5] + want[5:10], got[:5] + got[6:]))


def test_stage_properties_at_bench_scale(ctx):
    """Size-independent properties of the whole stage at a bench-like size (32 768 instances, planned natively, one
    device batch): chains are colinear (j, y and read intervals ascend as the solver allows), every MAM lies inside
    its read, the non-covered regions add up with the covered ones to the read length, values are finite and
    positive, re-running the staged batch reproduces the fetch bit for bit, and a seeded sample equals the oracle."""
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import synth
    w = synth.make_world(seed=51, n_genes=400, n_chr=4)
    idx = w.index()
    insts = synth.make_instances(w, 32768, seed=13, preset="ont_cdna")
    ix = native.NativeIndex(*idx)
    try:
        plan = ix.plan([(sol, read) for read, sol, _ in insts])
        stage.stage_batch(ctx, plan)
        stage.run_staged(ctx, 0.6)
        res1 = ctx.mam_fetch()
        stage.run_staged(ctx, 0.6)
        res2 = ctx.mam_fetch()
        for k in res1:
            assert np.array_equal(np.asarray(res1[k]), np.asarray(res2[k])), k
        out = stage.results_to_python(plan, res1)
    finally:
        ix.close()
    assert len(out) == len(insts)
    n_nonempty = 0
    for (read, sol, _), (gaps, value, mams) in zip(insts, out):
        if not mams:
            assert (gaps, value) == ([], -1)
            continue
        n_nonempty += 1
        assert np.isfinite(value) and value > 0
        js = [m.j for m in mams]
        assert js == sorted(set(js))
        assert all(m.y >= p.y for p, m in zip(mams[:-1], mams[1:]))          # sorted by y before chaining (:553)
        assert all(0 <= m.c <= m.d < len(read) for m in mams)
        covered = sum(m.d - m.c + 1 for m in mams)
        # gaps = [c0, c1 - d0 - 1, ..., len - d_last] (:579-592; the last term has no -1, overlaps make terms negative)
        assert sum(gaps) + covered == len(read) + 1
    assert n_nonempty > 0.9 * len(insts)
    for i in range(0, len(insts), 997):
        read, sol, _ = insts[i]
        assert _same(classify_port.classify(sol, *idx, read, None, 0.6), out[i])


def test_restaging_pairs_invalidates_the_mam_batch(ctx):
    """A MAM batch's candidate arrays are permuted for the pair batch it staged itself.  Staging other pairs in the
    same context (ultra_myers_stage, or ultra_myers_hw_batch as edlib_shim.align does) must make a repeated
    ultra_mam_run answer ULTRA_E_STATE instead of reading stale pair indices."""
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import lib, synth
    w = synth.make_world(seed=22, n_genes=10)
    insts = synth.make_instances(w, 40, seed=5)
    batch = stage.prepare_batch([(sol, read) for read, sol, _ in insts], *w.index())
    small = helpers.make_pairs(np.random.RandomState(1), 5, (9, 40), (50, 100))
    for restage in ("stage", "hw_batch"):
        stage.stage_batch(ctx, batch)
        stage.run_staged(ctx, 0.6)
        first = ctx.mam_fetch()
        if restage == "stage":
            ctx.myers_stage(*small)
        else:
            ctx.myers_hw_batch(*small)
        with pytest.raises(lib.UltraError) as err:
            ctx.mam_run(0.6)
        assert err.value.code == -6  # ULTRA_E_STATE
        stage.stage_batch(ctx, batch)   # staging again recovers
        stage.run_staged(ctx, 0.6)
        again = ctx.mam_fetch()
        for k in first:
            assert np.array_equal(np.asarray(first[k]), np.asarray(again[k])), k


def test_soft_masked_and_iupac_batches(ctx):
    """A batch with more than 16 distinct bytes - soft-masked (lower-case) segments, IUPAC codes and N in reads and
    segments - goes through the whole stage, host-planned and device-planned, and equals the oracle: the reference
    compares raw bytes and never upper-cases (uLTRA:43, create_augmented_gene.py:556)."""
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import synth
    w = synth.make_world(seed=31, n_genes=40)
    rng = np.random.RandomState(5)
    codes = "NRYKMSWBVDHnrykmswbvdh"

    def spoil(seq, p_mask, p_code):
        s = list(seq)
        if rng.rand() < p_mask and len(s) > 4:
            a = int(rng.randint(0, len(s) - 2)); b = int(rng.randint(a + 1, len(s)))
            s[a:b] = [c.lower() for c in s[a:b]]
        for i in range(len(s)):
            if rng.rand() < p_code:
                s[i] = codes[int(rng.randint(0, len(codes)))]
        return "".join(s)
    for d in (w.ref_segment_sequences, w.ref_flank_sequences):
        for k in list(d):
            d[k] = spoil(d[k], 0.3, 0.004)
    idx = w.index()
    insts = synth.make_instances(w, 160, seed=9, preset="ont_cdna", flank_start_prob=0.3, flank_end_prob=0.3)
    insts = [(spoil(read, 0.5, 0.003), sol, c) for read, sol, c in insts]
    distinct = set("".join(r for r, _, _ in insts)) | set("".join(w.ref_segment_sequences.values()))
    assert len(distinct) > 20
    pairs = [(sol, read) for read, sol, _ in insts]
    want = [classify_port.classify(sol, *idx, read, None, 0.6) for read, sol, _ in insts]
    got = stage.main_batch(pairs, *idx, min_acc=0.6, ctx=ctx)
    assert all(_same(a, b) for a, b in zip(want, got))
    assert sum(1 for g in got if g[2]) > 100
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    try:
        flat = native.flatten_instances(pairs)
        ctx.mam_stage_nams(dix, *flat)
        stage.run_staged(ctx, 0.6)
        dev = stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist())
    finally:
        dix.close(); ix.close()
    assert all(_same(a, b) for a, b in zip(want, dev))


def _hand_world():
    """A hand-built chromosome for the long call shapes (classify_read_with_mams.py:311-312, :374-377):
    flank [48000, 50000) | part [50000, 50900) of three exons | flank [50900, 52900) ... part [100000, 125000): one
    25 kb segment (a long last exon)."""
    from array import array
    from collections import defaultdict
    from tests import helpers
    rng = np.random.RandomState(17)
    key = lambda a, b: array("L", [1, a, b]).tobytes()
    genome = {}

    def seq(a, b):
        genome[(a, b)] = helpers.rand_seq(rng, b - a)
        return genome[(a, b)]
    segs = {key(50000, 50300): seq(50000, 50300), key(50300, 50600): seq(50300, 50600), key(50600, 50900): seq(50600, 50900),
            key(100000, 125000): seq(100000, 125000)}
    flanks = {key(48000, 50000): seq(48000, 50000), key(50900, 52900): seq(50900, 52900), key(98000, 100000): seq(98000, 100000)}
    parts = {key(50000, 50900): array("L", [1, 50000, 50300, 1, 50300, 50600, 1, 50600, 50900]), key(100000, 125000): array("L", [1, 100000, 125000])}
    seg_gene = {key(50000, 50300): {"g1"}, key(50300, 50600): {"g1"}, key(50600, 50900): {"g1"}, key(100000, 125000): {"g2"}}
    small = defaultdict(lambda: array("L"))
    return genome, (segs, flanks, parts, seg_gene, small)


def _nams(pieces, part_of):
    """pieces: [(ref start, read start, length)] exact seeds -> Mem list in reference order."""
    from ultra_amd.synth import Mem
    out = []
    for j, (x, c, n) in enumerate(sorted(pieces)):
        out.append(Mem(x, x + n - 1, c, c + n - 1, n, j, part_of(x)))
    return out


def test_long_call_shapes_at_stage_level(ctx):
    """Call shape (iii) - the read lies inside a segment at least 0.8 x its length (:374-377) - with 5-15 kb reads in a
    25 kb segment, and a 2 kb flank under a 2.2-2.6 kb read (shapes (i) and (iii) on the same candidate, partial flank
    pieces, :311-312), through the whole stage with the host and the device planner, against the oracle."""
    from oracle import classify_port
    from tests import helpers
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    genome, idx = _hand_world()
    rng = np.random.RandomState(23)

    def part_of(x):
        for a, b in ((48000, 50000), (50000, 50900), (50900, 52900), (98000, 100000), (100000, 125000)):
            if a <= x < b:
                return "1^%d^%d" % (a, b)
        raise AssertionError(x)

    def seeds_along(ref_start, read, truth_start, n_seeds, ln=18):
        """exact seeds can only be planted where the read still equals the genome: look for them"""
        flat = "".join(genome[k] for k in sorted(genome) if k[0] >= truth_start[0] and k[0] < truth_start[1])
        base = truth_start[0]
        out = []
        for _ in range(400):
            c = int(rng.randint(0, len(read) - ln))
            at = flat.find(read[c:c + ln])
            if at >= 0 and flat.find(read[c:c + ln], at + 1) < 0:
                x = base + at
                if part_of(x) == part_of(x + ln - 1):
                    out.append((x, c, ln))
            if len(out) >= n_seeds:
                break
        # colinear subset
        out.sort()
        keep, last_c = [], -1
        for x, c, n in out:
            if c > last_c:
                keep.append((x, c, n)); last_c = c + n
        return keep

    pairs = []
    long_seg = genome[(100000, 125000)]
    for ln in (5000, 7000, 9000, 12000, 15000, 15000, 6000, 11000):
        a = int(rng.randint(0, len(long_seg) - ln))
        read = helpers.mutate(rng, long_seg[a:a + ln], float(rng.choice([0.03, 0.08, 0.12])))
        nams = seeds_along(100000 + a, read, (100000, 125000), 12)
        assert len(nams) >= 2
        pairs.append((_nams(nams, part_of), read))
    # a read running from the 3' flank of the upstream gene into the long exon start: flank + contained shapes together
    span = genome[(98000, 100000)][-1500:] + long_seg[:6000]
    read = helpers.mutate(rng, span, 0.06)
    pairs.append((_nams(seeds_along(98500, read, (98000, 125000), 14), part_of), read))
    # 2 kb flank x 2.2-2.6 kb reads: the flank in front of the three-exon part, and the one behind it
    for lo, hi in ((48000, 50400), (47990 + 10, 50250), (50500, 52900), (50700, 52900)):
        lo = max(lo, 48000)
        ref = "".join(genome[k] for k in sorted(genome) if 48000 <= k[0] < 52900)
        piece = ref[lo - 48000:hi - 48000]
        for rate in (0.0, 0.05, 0.1):
            read = helpers.mutate(rng, piece, rate)
            nams = seeds_along(lo, read, (48000, 52900), 10)
            assert len(nams) >= 2
            pairs.append((_nams(nams, part_of), read))
    assert sum(len(r) >= 4500 for _, r in pairs) >= 9
    want = [classify_port.classify(sol, *idx, read, None, 0.6) for sol, read in pairs]
    assert sum(1 for w in want if w[2]) >= len(pairs) - 2
    labels = "".join(m.mam_id for w in want for m in w[2])
    assert "_full_flank" in labels or "_partial_flank" in labels
    assert any(m.c == 0 and m.d == len(read) - 1 for w, (_, read) in zip(want, pairs) for m in w[2] if len(read) >= 4500)   # shape (iii) taken
    got = stage.main_batch(pairs, *idx, min_acc=0.6, ctx=ctx)
    assert all(_same(a, b) for a, b in zip(want, got))
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    try:
        flat = native.flatten_instances(pairs)
        ctx.mam_stage_nams(dix, *flat)
        stage.run_staged(ctx, 0.6)
        dev = stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist())
    finally:
        dix.close(); ix.close()
    assert all(_same(a, b) for a, b in zip(want, dev))
