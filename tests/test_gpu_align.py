"""GPU: the batched front end (ultra_amd.align) - seeds -> instances -> ONE NAM chaining launch -> location
selection -> ONE MAM stage - against the reference's per-read loop recorded in tests/golden/align_golden.json and,
for the MAM stage of the selected locations, against the oracle."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_golden.json")


def rows(mems):
    return [[m.x, m.y, m.c, m.d, m.val, m.j, m.exon_part_id] for m in mems]


def test_location_selection_matches_reference_loop(ctx):
    from ultra_amd import align
    gold = json.load(open(GOLD))
    intron = {int(k): v for k, v in gold["max_intron_chr"].items()}
    records = [(hits, hits_rc) for _, hits, _, hits_rc in gold["parsed"]]
    for run in gold["runs"]:
        p = run["params"]
        got = align.chain_reads_batch(records, intron, p["max_global_intron"], p["dropoff"], p["max_loc"], ctx=ctx)
        assert len(got) == len(run["reads"])
        n_multi = 0
        for locs, want in zip(got, run["reads"]):
            assert [[c, rows(sol), score, rc] for c, sol, score, rc in locs] == want["selected"]
            n_multi += len(locs) > 1
        assert n_multi > 0


def test_front_end_to_mam_stage(ctx):
    """Seeds of simulated reads through chain_reads_batch + mam_stage_batch; every (read, location) result equals
    the oracle's classify() on the same chain, and a read's repeated MAM solutions are dropped."""
    import numpy as np
    from oracle import classify_port
    from ultra_amd import align, seed_wrapper, synth
    world = synth.make_world(seed=77, n_genes=40, n_chr=2)
    rng = np.random.RandomState(3)
    sims = [synth.simulate_instance(world, rng, "ont_cdna") for _ in range(40)]
    records = [([seed_wrapper.format_hit(m) for m in sol], []) for _, sol, _ in sims]
    intron = {c: 100000 for c in {p[0] for p in world.part_seq}}
    locs = align.chain_reads_batch(records, intron, 1200000, 0.95, 5, ctx=ctx)
    reads = [read for read, _, _ in sims]
    got = align.mam_stage_batch(reads, locs, world.index(), 0.6, ctx=ctx)
    n_loc = 0
    for r, (read, sol, chr_id) in enumerate(sims):
        assert locs[r] and locs[r][0][0] == chr_id
        want, seen = [], set()
        for (c, chain, score, is_rc) in locs[r]:
            res = classify_port.classify(chain, *world.index(), read, None, 0.6)
            if res[2] in seen:
                continue
            seen.add(res[2])
            want.append((c, is_rc, score) + tuple(res))
        assert got[r] == want
        n_loc += len(want)
    assert n_loc >= len(sims)
