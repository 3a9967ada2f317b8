"""GPU: the batch pipeline (ultra_pipe) and the restaged align_single / pc front end.

ultra_pipe must return, batch by batch and in submission order, exactly what the synchronous path returns, whatever the
depth; align_single must not care how the reads are grouped into GPU batches."""
import os
import queue
import struct
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (a[0] == b[0] and struct.pack("d", float(a[1])) == struct.pack("d", float(b[1]))
            and tuple(tuple(m) for m in a[2]) == tuple(tuple(m) for m in b[2]))


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipe_matches_synchronous_path(ctx, depth):
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native
    from ultra_amd import lib, synth
    w = synth.make_world(seed=71, n_genes=120, n_chr=3, overlap_frac=0.3, exon_mean=12.0, max_exons=40)
    idx = w.index()
    flat = synth.FlatWorld(w)
    rs = synth.make_reads_native(flat, 6000, seed=3, preset="ont_cdna", threads=4)
    ix = native.NativeIndex(*idx)
    dix = native.DeviceIndex(ix)
    subs = [rs.slice(a, b) for a, b in ((0, 2500), (2500, 2600), (2600, 6000), (0, 1))]
    want = []
    try:
        for s in subs:
            ctx.mam_stage_nams(dix, s.nam_off, s.nam_part, s.nam_x, s.nam_y, s.reads, s.read_off)
            ctx.mam_run(0.6)
            want.append(stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(s.read_off).tolist()))
        pipe = lib.Pipe(ix, 0, depth=depth)
        try:
            got = []
            order = [0, 1, 2, 3, 2, 0]
            sent = 0
            while len(got) < len(order):
                while sent < len(order) and pipe.pending() < depth:
                    s = subs[order[sent]]
                    pipe.submit(s.nam_off, s.nam_part, s.nam_x, s.nam_y, s.reads, s.read_off, 0.6)
                    sent += 1
                res = pipe.wait()
                s = subs[order[len(got)]]
                got.append(stage.results_from_names(res, np.diff(s.read_off).tolist()))
                t = pipe.timing()
                assert t["run_ms"] > 0 and t["cells"] > 0
                pipe.release()
            with pytest.raises(lib.UltraError):
                pipe.wait()                         # nothing submitted
        finally:
            pipe.close()
    finally:
        dix.close(); ix.close()
    for k, g in zip(order, got):
        assert len(g) == len(want[k])
        assert all(_same(a, b) for a, b in zip(g, want[k]))
    for i in range(0, 2500, 97):                    # and the synchronous path equals the oracle
        read, sol, _ = subs[0].instance(i)
        assert _same(classify_port.classify(sol, *idx, read, None, 0.6), want[0][i])


def _stub_downstream():
    """Stand-ins for the stages behind the MAM stage (they stay the reference's code in a real installation): enough
    structure to exercise every branch of the restaged loop, deterministic, and cheap."""
    from ultra_amd import align

    def find_exons(chr_id, mam_solution, ref_exon_sequences, ref_segment_sequences, ref_flank_sequences, splice_pairs):
        exons = [(m.x, m.y, m.c, m.d, m.ref_chr_id) for m in mam_solution]
        seq = "".join(ref_segment_sequences.get(align._array('L', [m.ref_chr_id, m.x, m.y]).tobytes(), "") for m in mam_solution)
        pred = [(m.x, m.y) for m in mam_solution]
        splices = [(a[1], b[0]) for a, b in zip(pred[:-1], pred[1:])]
        return exons, seq, pred, splices, sum(m.d - m.c + 1 for m in mam_solution)

    def classify(chr_id, splices, *rest):
        return ("NO_SPLICE", "") if not splices else ("FSM" if len(splices) % 2 else "NIC_novel", "tr%d" % len(splices))

    def exact_alignment(read_seq, ref_seq, exons_len):
        n = min(len(read_seq), len(ref_seq))
        score = 2 * sum(1 for a, b in zip(read_seq[:n], ref_seq[:n]) if a == b) - abs(len(read_seq) - len(ref_seq))
        return read_seq[:n], ref_seq[:n], score

    def sam_entry(*a, **kw):
        return "\t".join(str(v) for v in a) + "\t" + str(sorted(kw.items())) + "\n"
    return align.Downstream(find_exons, classify, exact_alignment, sam_entry)


def test_align_single_is_independent_of_gpu_batching(ctx, tmp_path):
    from ultra_amd import align, seed_wrapper, synth
    world = synth.make_world(seed=77, n_genes=40, n_chr=2)
    rng = np.random.RandomState(3)
    sims = [synth.simulate_instance(world, rng, "ont_cdna") for _ in range(90)]
    chr_ids = sorted({p[0] for p in world.part_seq})
    data = (world.ref_segment_sequences, world.ref_flank_sequences, {}, {}, {c: {} for c in chr_ids}, {c: set() for c in chr_ids},
            world.parts_to_segments, world.segment_to_gene, world.gene_to_small_segments, {c: 100000 for c in chr_ids},
            {}, {str(c): c for c in chr_ids}, {c: "chr%d" % c for c in chr_ids})
    state = align.AlignState(data, ctx=ctx, device=0)
    records = []
    for i, (read, sol, _) in enumerate(sims):
        hits = [seed_wrapper.format_hit(m) for m in sol]
        if i % 11 == 0:
            hits = []                                                   # a read without seeds -> 'unaligned'
        qual = None if i % 2 else "I" * len(read)
        records.append(["read%d" % i, read + ("A" * 20 if i % 5 == 0 else ""), (qual + "I" * 20 if qual and i % 5 == 0 else qual), hits, []])
    args = types.SimpleNamespace(outfolder=str(tmp_path), index="", reduce_read_ployA=8, max_intron=1200000, dropoff=0.95, max_loc=5,
                                 min_acc=0.6, alignment_threshold=0.5, non_covered_cutoff=15, downstream=_stub_downstream(),
                                 ultra_state=state)
    try:
        runs = []
        for per_queue_batch, gpu_batch in ((1000, 65536), (7, 20), (1, 1)):
            qin, qout, qcnt = queue.Queue(), queue.Queue(), queue.Queue()
            for b in range(0, len(records), per_queue_batch):
                qin.put([b // per_queue_batch + 1, records[b:b + per_queue_batch]])
            qin.put(None)
            args.gpu_batch_reads = gpu_batch
            align.align_single(0, qin, qout, qcnt, args)
            assert qin.get() is None                                    # the sentinel stays for other consumers
            rows = []
            n_lists = 0
            while not qout.empty():
                rows.extend(qout.get()); n_lists += 1
            assert n_lists == (len(records) + per_queue_batch - 1) // per_queue_batch
            runs.append((rows, qcnt.get()))
    finally:
        state.close()
    assert runs[0][0] == runs[1][0] == runs[2][0]
    assert runs[0][1][1:] == runs[1][1][1:] == runs[2][1][1:]
    assert abs(runs[0][1][0] - runs[2][1][0]) < 1e-9
    rows = runs[0][0]
    assert len(rows) >= len(records)
    assert sum("unaligned" in r for r in rows) >= len(records) // 11
    fields = [r.split("\t") for r in rows]                              # read_id, seq, qual, ref, classification, ..., is_rc, is_secondary, mapq
    primaries = [f for f in fields if f[4] != "unaligned" and f[10] == "False"]
    assert sum(runs[0][1][1:]) == len(primaries) > len(records) // 4     # one classification count per aligned read
    assert all(f[11] in ("0", "60") for f in primaries)


def test_one_read_outside_the_envelope_does_not_take_the_batch_down(ctx, tmp_path):
    """A read the GPU path cannot take (here: more distinct bytes than the 36 a batch may hold) fails the launch it is
    part of; align_single then isolates it, reports it unaligned with a line in uLTRA_batch_N.stderr, and every other
    read of the batch gets exactly the rows it gets without it.  A 40 kb read contained in a 50 kb exon (queries
    beyond 32768 rows, classify_read_with_mams.py:374-377) is NOT such a read: it is aligned."""
    from ultra_amd import align, seed_wrapper, synth
    world = synth.make_world(seed=78, n_genes=30, n_chr=2)
    rng = np.random.RandomState(4)
    # one very long exon, hit by a very long read
    chr_id = 0
    big = (chr_id, 9000000, 9050000)
    big_seq = synth._rand_seq(rng, big[2] - big[1])
    key = synth.key_of(*big)
    world.part_seq[big] = big_seq
    world.ref_segment_sequences[key] = big_seq
    world.parts_to_segments[key] = align._array('L', list(big))
    world.segment_to_gene[key] = {"gene_big"}
    world.gene_to_small_segments["gene_big"] = align._array('L')
    sims = [synth.simulate_instance(world, rng, "ont_cdna") for _ in range(60)]
    chr_ids = sorted({p[0] for p in world.part_seq})
    data = (world.ref_segment_sequences, world.ref_flank_sequences, {}, {}, {c: {} for c in chr_ids}, {c: set() for c in chr_ids},
            world.parts_to_segments, world.segment_to_gene, world.gene_to_small_segments, {c: 100000 for c in chr_ids},
            {}, {str(c): c for c in chr_ids}, {c: "chr%d" % c for c in chr_ids})
    state = align.AlignState(data, ctx=ctx, device=0)
    records = [["read%d" % i, read, None, [seed_wrapper.format_hit(m) for m in sol], []] for i, (read, sol, _) in enumerate(sims)]
    # the long read: 40 kb of the exon with a few edits, seeds = its exact stretches (one NAM per 500 bases is plenty)
    from tests import helpers
    # (substitutions only, from the exon's first base: the stub base-level alignment of this test compares position by position)
    long_read = helpers.mutate(rng, big_seq[:40000], 0.01, mix=(0.0, 1.0, 0.0))
    hits = []
    for c in range(0, len(long_read) - 40, 2000):
        piece = long_read[c:c + 25]
        at = big_seq.find(piece)
        if at >= 0 and big_seq.find(piece, at + 1) < 0:
            hits.append("%d^%d^%d\t%d\t%d\t%d\n" % (chr_id, big[1], big[2], at + 1, c + 1, 25))
    assert len(hits) > 10
    long_rec = ["long_read", long_read, None, hits, []]
    # the read no batch can hold: its own 40 distinct bytes on top of the index's
    weird = sims[7][0][:200] + "".join(chr(c) for c in range(48, 88)) + sims[7][0][200:]
    weird_rec = ["weird_read", weird, None, records[7][3], []]
    args = types.SimpleNamespace(outfolder=str(tmp_path), index="", reduce_read_ployA=8, max_intron=1200000, dropoff=0.95, max_loc=5,
                                 min_acc=0.6, alignment_threshold=0.5, non_covered_cutoff=15, downstream=_stub_downstream(),
                                 ultra_state=state, gpu_batch_reads=65536)

    def run(recs):
        qin, qout, qcnt = queue.Queue(), queue.Queue(), queue.Queue()
        qin.put([1, recs]); qin.put(None)
        align.align_single(0, qin, qout, qcnt, args)
        return qout.get(), qcnt.get()
    try:
        clean_rows, clean_counts = run(records + [long_rec])
        log_clean = open(os.path.join(str(tmp_path), "uLTRA_batch_0.stderr")).read()
        mixed = records[:20] + [weird_rec] + records[20:] + [long_rec]
        rows, counts = run(mixed)
        log = open(os.path.join(str(tmp_path), "uLTRA_batch_0.stderr")).read()
    finally:
        state.close()
    assert "weird_read" not in log_clean and "weird_read" in log and "unaligned" in log
    weird_rows = [r for r in rows if r.startswith("weird_read\t")]
    assert len(weird_rows) == 1 and "unaligned" in weird_rows[0]
    assert [r for r in rows if not r.startswith("weird_read\t")] == clean_rows
    assert counts[1:] == clean_counts[1:]
    long_rows = [r for r in clean_rows if r.startswith("long_read\t")]
    assert long_rows and "unaligned" not in long_rows[0] and "9000000" in long_rows[0]


@pytest.mark.parametrize("gpu_batch", [65536, 7])
def test_align_single_equals_the_reference_on_sirv(ctx, tmp_path, gpu_batch):
    """tests/golden/align_single_golden.json: the reference's own align_single (modules/align.py:392-560) run on the
    SIRV annotation, with a transcript of every call it made into find_exons / classify_alignment2 / get_exact_alignment
    / sam_output.  Here the same reads go through ultra_amd.align.align_single on the GPU; the downstream stages are
    replayed from the transcript (a call the reference did not make fails the lookup), the base-level alignments come
    from the GPU batch and must equal the recorded ones, and the SAM rows and classification counts must be the
    reference's."""
    import json
    from tests import helpers
    from ultra_amd import align, indexer
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "align_single_golden.json")) as f:
        gold = json.load(f)
    sirv = os.path.join(here, "golden", "sirv")
    ann = indexer.build_index(os.path.join(sirv, "SIRV_genes_C_170612a.gtf"), os.path.join(sirv, "SIRV_genes.fasta"), **gold["index"])
    calls = gold["calls"]
    used = {k: set() for k in calls}

    def replay(name, key):
        d = helpers.digest(key)
        assert d in calls[name], "%s was called with arguments the reference never passed" % name
        used[name].add(d)
        return helpers.unjsonable(calls[name][d])

    def find_exons(chr_id, mam_solution, *tables):
        return replay("find_exons", [chr_id, helpers.mam_key(mam_solution)])

    def classify(chr_id, predicted_splices, *tables):
        return replay("classify", [chr_id, predicted_splices])

    def exact_batch(jobs):
        got = align.get_exact_alignment_batch(jobs, ctx=ctx)
        for job, g in zip(jobs, got):
            want = replay("exact_alignment", list(job))
            assert tuple(g) == tuple(want)
        return got

    def sam_entry(*a, **kw):
        return replay("sam_entry", [list(a), kw])

    state = align.AlignState(indexer.align_data(ann), ctx=ctx, device=0)
    try:
        for run in gold["runs"]:
            args = types.SimpleNamespace(outfolder=str(tmp_path), index="", gpu_batch_reads=gpu_batch, ultra_state=state,
                                         downstream=align.Downstream(find_exons, classify, None, sam_entry, exact_alignment_batch=exact_batch),
                                         **run["args"])
            qin, qout, qcnt = queue.Queue(), queue.Queue(), queue.Queue()
            for b in range(0, len(gold["records"]), 20):
                qin.put([b // 20 + 1, gold["records"][b:b + 20]])
            qin.put(None)
            align.align_single(0, qin, qout, qcnt, args)
            rows = []
            while not qout.empty():
                rows.append(qout.get())
            assert rows == run["rows"], run["name"]
            counts, want = qcnt.get(), helpers.unjsonable(run["classification_list"])
            assert counts[1:] == want[1:]
            assert abs(counts[0] - want[0]) < 1e-9          # a sum of per-read coverages, added in the same order
    finally:
        state.close()
    # every recorded SAM row and alignment was needed by one of the two runs
    assert used["sam_entry"] == set(calls["sam_entry"]) and used["exact_alignment"] == set(calls["exact_alignment"])


@pytest.mark.parametrize("per_gpu", [1, 2])
def test_pc_main_runs_the_reference_protocol(ctx, tmp_path, per_gpu):
    """ultra_amd.pc.main(reads, seeds, outfile, args) as uLTRA:382 calls it: a producer zipping the reads file with the
    seed file into 1000-read queue batches, one (or args.consumers_per_gpu) align_single consumer PROCESS per GPU reading the index pickles from
    args.outfolder, SAM rows appended to the output file, the 8-entry totals returned.  Same rows as align_single run
    in this process on the same records (base-level alignment on the GPU in both)."""
    from tests import helpers
    from ultra_amd import align, hwq, indexer, pc, seed_wrapper, seeder
    here = os.path.dirname(os.path.abspath(__file__))
    sirv = os.path.join(here, "golden", "sirv")
    out = str(tmp_path)
    ann = indexer.build_index(os.path.join(sirv, "SIRV_genes_C_170612a.gtf"), os.path.join(sirv, "SIRV_genes.fasta"), out)
    refs = indexer.read_fasta(os.path.join(sirv, "SIRV_genes.fasta"))
    rng = np.random.RandomState(5)
    reads = helpers.simulate_transcript_reads(indexer.read_gtf(os.path.join(sirv, "SIRV_genes_C_170612a.gtf")), refs, rng, 2300)
    reads_path, seeds_path, sam_path = os.path.join(out, "reads.fq"), os.path.join(out, "seeds.txt.gz"), os.path.join(out, "reads.sam")
    with open(reads_path, "w") as fh:
        for acc, seq, qual in reads:
            fh.write("@%s\n%s\n+\n%s\n" % (acc, seq, qual) if qual else ">%s\n%s\n" % (acc, seq))
    seeder.write_seed_file(seeds_path, [(a, s) for a, s, _ in reads], ann["ref_part_sequences"], min_len=15)
    open(sam_path, "w").close()
    down = align.Downstream(helpers.stub_find_exons, helpers.stub_classify, None, helpers.stub_sam_entry)
    args = types.SimpleNamespace(outfolder=out, index=out, reduce_read_ployA=20, max_intron=1200000, dropoff=0.95, max_loc=5, min_acc=0.6,
                                 alignment_threshold=0.5, non_covered_cutoff=15, gpus=1, gpu_batch_reads=1500, downstream=down,
                                 consumers_per_gpu=per_gpu)
    totals = pc.main(reads_path, seeds_path, sam_path, args)
    # the consumers are spawned children: they must run with the benched queue count, not HIP's default of 4
    for w in range(per_gpu):
        # the benched queue count, or the process's share of what one GPU schedules at once (two consumers: 11 each)
        want_q = hwq.DEFAULT if per_gpu == 1 else min(hwq.DEFAULT, hwq.GPU_BUDGET // per_gpu)
        assert "GPU_MAX_HW_QUEUES=%d," % want_q in open(os.path.join(out, "uLTRA_batch_%d.stderr" % w)).read()
    rows = open(sam_path).read().splitlines(True)
    # the same records through align_single in this process
    with open(reads_path) as fh:
        records = [[acc, seq, qual, hits, hits_rc] for (acc, (seq, qual)), (_, hits, _, hits_rc) in zip(pc.readfq(fh), seed_wrapper.read_seeds(seeds_path))]
    assert [r[0] for r in records] == [a for a, _, _ in reads] and [r[2] for r in records] == [q for _, _, q in reads]
    state = align.AlignState(indexer.align_data(ann), ctx=ctx, device=0)
    try:
        args2 = types.SimpleNamespace(**dict(vars(args), ultra_state=state, gpu_batch_reads=65536))
        qin, qout, qcnt = queue.Queue(), queue.Queue(), queue.Queue()
        for b in range(0, len(records), 1000):
            qin.put([b // 1000 + 1, records[b:b + 1000]])
        qin.put(None)
        align.align_single(0, qin, qout, qcnt, args2)
        want = []
        while not qout.empty():
            want.extend(qout.get())
        counts = qcnt.get()
    finally:
        state.close()
    assert len(rows) == len(want) >= len(reads)
    assert sorted(rows) == sorted(want)                 # batches come back in completion order
    assert totals[1:] == counts[1:] and abs(totals[0] - counts[0]) < 1e-6
    assert sum(totals[1:]) > len(reads) // 2
