"""GPU: the native front end (ultra_front_run: hit lines -> instances -> ONE NAM chaining launch -> location selection ->
ONE MAM stage) against the reference's per-read loop recorded in tests/golden/align_golden.json, against the Python
shims of ultra_amd.align (themselves pinned by that file) and, for the MAM stage, against the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_golden.json")


def rows(mems):
    return [[m.x, m.y, m.c, m.d, m.val, m.j, m.exon_part_id] for m in mems]


def test_location_selection_matches_reference_loop(ctx):
    from ultra_amd import front
    gold = json.load(open(GOLD))
    intron = {int(k): v for k, v in gold["max_intron_chr"].items()}
    recs = [(acc, read, None, hits, hits_rc) for (acc, hits, _, hits_rc), read in zip(gold["parsed"], gold["reads"])]
    pb = front.PackedBatch.from_records(recs)
    fr = front.Front(ctx, max_intron_chr=intron)
    try:
        for run in gold["runs"]:
            p = run["params"]
            res = fr.run(pb, p["max_global_intron"], p["dropoff"], p["max_loc"], 20, 0.6, with_mam_stage=False)
            assert res.n_reads == len(run["reads"])
            n_multi = 0
            for r, want in enumerate(run["reads"]):
                got = [[int(res.loc_chr[l]), rows(res.solution(l)), int(res.loc_score[l]), bool(res.loc_rc[l])]
                       for l in range(int(res.loc_off[r]), int(res.loc_off[r + 1]))]
                assert got == want["selected"], r
                n_multi += len(got) > 1
            assert n_multi > 0
    finally:
        fr.close()


def test_front_to_mam_stage_equals_python_path_and_oracle(ctx):
    """Simulated reads on both strands, some with poly-A tails, some without seeds: locations, last intron limits,
    trimmed reads and every (read, location) MAM result equal what ultra_amd.align's per-read Python produces, and the
    MAM results equal the oracle's classify() on the same chain."""
    from oracle import classify_port
    from ultra_amd import align, front, index as native, seed_wrapper, synth
    world = synth.make_world(seed=79, n_genes=60, n_chr=3, overlap_frac=0.2)
    rng = np.random.RandomState(5)
    sims = [synth.simulate_instance(world, rng, "ont_cdna") for _ in range(120)]
    recs = []
    for i, (read, sol, _) in enumerate(sims):
        hits = [seed_wrapper.format_hit(m) for m in sol]
        tail = "A" * 30 if i % 4 == 0 else ""
        if i % 3 == 1:      # the read as sequenced is the reverse complement: its seeds arrive in the "Reverse" record
            recs.append(("r%d" % i, align.reverse_complement(read + tail), None, [], hits))
        elif i % 10 == 5:
            recs.append(("r%d" % i, read, None, [], []))
        else:
            # decoy seeds on another chromosome in the forward record too
            other = sims[(i + 7) % len(sims)][1][:3]
            recs.append(("r%d" % i, read + tail, None, hits + [seed_wrapper.format_hit(m) for m in other], []))
    intron = {c: 100000 for c in {p[0] for p in world.part_seq}}
    ix = native.NativeIndex(*world.index())
    dix = native.DeviceIndex(ix)
    fr = front.Front(ctx, dix, ix, intron)
    try:
        res = fr.run(front.PackedBatch.from_records(recs), 1200000, 0.95, 5, 8, 0.6)
        limits = []
        want_locs = align.chain_reads_batch([(r[3], r[4]) for r in recs], intron, 1200000, 0.95, 5, ctx=ctx, last_limits=limits)
        assert [None if v < 0 else int(v) for v in res.last_limit] == limits
        mods = [align.remove_read_polyA_ends(r[1], None, 8, 1)[0] for r in recs]
        assert [res.read_mod_seq(r) for r in range(len(recs))] == mods
        n_loc = n_rc = 0
        for r, locs in enumerate(want_locs):
            a, b = int(res.loc_off[r]), int(res.loc_off[r + 1])
            assert b - a == len(locs)
            for l, (chr_id, sol, score, is_rc) in zip(range(a, b), locs):
                assert (int(res.loc_chr[l]), int(res.loc_score[l]), bool(res.loc_rc[l])) == (chr_id, score, is_rc)
                assert rows(res.solution(l)) == rows(sol)
                read = align.reverse_complement(mods[r]) if is_rc else mods[r]
                assert res.loc_read_seq(l) == read
                want = classify_port.classify(sol, *world.index(), read, None, 0.6)
                got = res.mam(l)
                assert got[0] == want[0] and float(got[1]).hex() == float(want[1]).hex()
                assert tuple(tuple(m) for m in got[2]) == tuple(tuple(m) for m in want[2])
                n_loc += 1; n_rc += is_rc
        assert n_loc >= 100 and n_rc >= 30 and res.n_loc == n_loc
        # an empty batch, and a batch of reads without any seed
        e = fr.run(front.PackedBatch.from_records([]), 1200000, 0.95, 5, 8, 0.6)
        assert e.n_loc == 0 and e.n_reads == 0
        e = fr.run(front.PackedBatch.from_records([("x", "ACGTACGT", None, [], [])]), 1200000, 0.95, 5, 8, 0.6)
        assert e.n_loc == 0 and e.last_limit.tolist() == [-1]
        # a character reverse_complement does not know, on a read whose location is on the reverse strand
        bad = list(recs[1]); bad[1] = bad[1][:50] + "-" + bad[1][51:]
        from ultra_amd import lib
        with pytest.raises(lib.UltraError) as err:
            fr.run(front.PackedBatch.from_records([recs[0], tuple(bad)]), 1200000, 0.95, 5, 8, 0.6)
        assert "read 1" in str(err.value) and "reverse_complement" in str(err.value)
    finally:
        fr.close(); dix.close(); ix.close()
