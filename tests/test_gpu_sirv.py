"""BASELINE.json configs[0], the bundled plumbing run: the SIRV annotation and the 4 reads of /root/reference/test/
(data fixtures under tests/golden/sirv/) travel index -> seeds -> NAM chaining -> MAM stage on the GPU, and every
stage is compared with the CPU oracle.  The index comes from ultra_amd.indexer (pinned against the reference in
tests/test_indexer.py), the seeds from the exact-match stand-in for namfinder (ultra_amd.seeder: the namfinder binary
is not part of this image, so the seeds are not namfinder's; every stage after them is fed the same seeds on both
sides)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIRV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sirv")


class OracleChainer(object):
    """Stands in for a Context in colinear_solver.read_coverage_batch: the DP vectors of the CPU oracle."""

    def chain_nams_batch(self, off, y, c, d, max_introns, solver=0):
        from oracle import cport
        y, c, d = np.asarray(y, np.int32), np.asarray(c, np.int32), np.asarray(d, np.int32)
        C, tr = [], []
        for i in range(len(off) - 1):
            lo, hi = int(off[i]), int(off[i + 1])
            if solver == 1 or (solver == 0 and hi - lo < 90):
                wc, wt = cport.nam_quadratic(y[lo:hi], c[lo:hi], d[lo:hi], int(max_introns[i]))
            else:
                wc, wt = cport.nam_nlogn(c[lo:hi], d[lo:hi])
            C += list(wc); tr += list(wt)
        return np.asarray(C, np.int64), np.asarray(tr, np.int64), None


@pytest.fixture(scope="module")
def sirv(tmp_path_factory):
    from ultra_amd import indexer, pc, seed_wrapper, seeder
    ann = indexer.build_index(os.path.join(SIRV, "SIRV_genes_C_170612a.gtf"), os.path.join(SIRV, "SIRV_genes.fasta"))
    with open(os.path.join(SIRV, "reads.fa")) as fh:
        reads = [(acc, seq) for acc, (seq, _) in pc.readfq(fh)]
    path = str(tmp_path_factory.mktemp("sirv") / "seeds.txt.gz")
    n_hits = seeder.write_seed_file(path, reads, ann["ref_part_sequences"], min_len=15)
    records = list(seed_wrapper.read_seeds(path))
    return ann, reads, records, n_hits


def test_sirv_reads_through_chaining_and_mam_stage(ctx, sirv):
    from oracle import classify_port
    from ultra_amd import align
    ann, reads, records, n_hits = sirv
    assert [r[0] for r in reads] == ["read1", "read2", "read3", "read4"] == [rec[0] for rec in records]
    assert n_hits > 20
    seeds = [(hits, hits_rc) for _, hits, _, hits_rc in records]
    trimmed = [align.remove_read_polyA_ends(seq, None, 20, 5)[0] for _, seq in reads]
    args = (ann["max_intron_chr"], 1200000, 0.95, 5)

    got = align.chain_reads_batch(seeds, *args, ctx=ctx)
    want = align.chain_reads_batch(seeds, *args, ctx=OracleChainer())
    assert got == want
    assert all(len(locs) >= 1 for locs in got)           # every bundled read finds a location

    five = tuple(ann[k] for k in ("ref_segment_sequences", "ref_flank_sequences", "parts_to_segments", "segment_to_gene", "gene_to_small_segments"))
    for min_acc in (0.5, 0.6, 0.8):
        per_read = align.mam_stage_batch(trimmed, got, five, min_acc, ctx=ctx)
        n = 0
        for r, locs in enumerate(got):
            seen, expect = set(), []
            for chr_id, sol, score, is_rc in locs:
                seq = align.reverse_complement(trimmed[r]) if is_rc else trimmed[r]
                non_covered, value, mam_sol = classify_port.classify(sol, *five, seq, None, min_acc)
                if mam_sol in seen:
                    continue
                seen.add(mam_sol)
                expect.append((chr_id, is_rc, score, non_covered, value, mam_sol))
            assert len(per_read[r]) == len(expect)
            for g, w in zip(per_read[r], expect):
                assert g[:3] == w[:3]
                assert list(g[3]) == list(w[3]) and g[4] == w[4]
                assert [tuple(m) for m in g[5]] == [tuple(m) for m in w[5]]
                n += len(g[5])
        assert n > 0                                     # and is covered by at least one MAM


def test_sirv_reads_land_on_their_genes(ctx, sirv):
    """The read names of the fixture carry no truth, so the check is structural: each read's best location chains
    seeds of ONE SIRV contig, covers most of the read, and the MAM solution lies on that contig."""
    from ultra_amd import align
    ann, reads, records, _ = sirv
    seeds = [(hits, hits_rc) for _, hits, _, hits_rc in records]
    trimmed = [align.remove_read_polyA_ends(seq, None, 20, 5)[0] for _, seq in reads]
    locs = align.chain_reads_batch(seeds, ann["max_intron_chr"], 1200000, 0.95, 5, ctx=ctx)
    five = tuple(ann[k] for k in ("ref_segment_sequences", "ref_flank_sequences", "parts_to_segments", "segment_to_gene", "gene_to_small_segments"))
    per_read = align.mam_stage_batch(trimmed, locs, five, 0.6, ctx=ctx)
    for r, out in enumerate(per_read):
        chr_id, is_rc, score, non_covered, value, mam_sol = out[0]
        assert score >= 0.5 * len(trimmed[r])
        assert mam_sol and all(m.ref_chr_id == chr_id for m in mam_sol)
        assert value >= 0.5 * len(trimmed[r])


def test_simulated_sirv_reads_host_and_device_planner_vs_oracle(ctx, tmp_path):
    """400 reads simulated from the annotated SIRV transcripts.  A real annotation has what the synthetic worlds lack:
    segments that start at the same coordinate (alternative 3' ends, the widened segments of short intervals), where
    the reference's candidate order is the iteration order of a Python set (csrc/pyset.h).  Both planners - host
    (ultra_plan_create) and device (ultra_mam_stage_nams) - must reproduce the oracle, which runs real Python sets."""
    from oracle import classify_port
    from tests import helpers
    from ultra_amd import align, indexer, seed_wrapper, seeder
    ann = indexer.build_index(os.path.join(SIRV, "SIRV_genes_C_170612a.gtf"), os.path.join(SIRV, "SIRV_genes.fasta"))
    refs = indexer.read_fasta(os.path.join(SIRV, "SIRV_genes.fasta"))
    rng = np.random.RandomState(99)
    reads = helpers.simulate_transcript_reads(indexer.read_gtf(os.path.join(SIRV, "SIRV_genes_C_170612a.gtf")), refs, rng, 400)
    path = str(tmp_path / "seeds.txt.gz")
    seeder.write_seed_file(path, [(a, s) for a, s, _ in reads], ann["ref_part_sequences"], min_len=15)
    seeds = [(hits, hits_rc) for _, hits, _, hits_rc in seed_wrapper.read_seeds(path)]
    trimmed = [align.remove_read_polyA_ends(seq, None, 20, 1)[0] for _, seq, _ in reads]
    locs = align.chain_reads_batch(seeds, ann["max_intron_chr"], 1200000, 0.95, 5, ctx=ctx)
    instances = []
    for r, per_read in enumerate(locs):
        for chr_id, sol, score, is_rc in per_read:
            instances.append((sol, align.reverse_complement(trimmed[r]) if is_rc else trimmed[r]))
    assert len(instances) > 300
    state = align.AlignState(indexer.align_data(ann), ctx=ctx, device=0)
    try:
        assert state.dindex is not None
        five = state.five()
        want = [classify_port.classify(sol, *five, read, None, 0.6) for sol, read in instances]
        ties = 0
        for sol, _ in instances:       # the case this test is for does occur
            hit = set()
            for nam in sol:
                key = align._array('L', [int(v) for v in nam.exon_part_id.split('^')]).tobytes()
                flat = ann["parts_to_segments"].get(key)
                if flat is not None:
                    hit.update((flat[i + 1], flat[i + 2]) for i in range(0, len(flat), 3) if not (flat[i + 2] < nam.x or nam.y < flat[i + 1]))
            starts = [s for s, _ in hit]
            ties += len(starts) != len(set(starts))
        assert ties > 20
        from ultra_amd import classify_read_with_mams as stage
        host = stage.main_batch(instances, *five, min_acc=0.6, ctx=ctx, index=state.native)
        dev = state.mam_stage(instances, 0.6)
        for name, got in (("host planner", host), ("device planner", dev)):
            bad = [i for i, (g, w) in enumerate(zip(got, want))
                   if list(g[0]) != list(w[0]) or g[1] != w[1] or helpers.mam_key(g[2]) != helpers.mam_key(w[2])]
            assert not bad, (name, len(bad), bad[:5])
    finally:
        state.close()
