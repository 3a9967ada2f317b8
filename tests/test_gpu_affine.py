"""GPU: ultra_sg_affine_batch (csrc/affine.hip) against the oracle (oracle/affine_port.c):
score, end cell and edit script bit for bit, through the C-ABI.  What is pinned about the oracle itself: its score,
by a plain DP (tests/test_oracle_affine.py); parasail's own tie-breaking is parity-unpinned."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _batch(rng, n_pairs, qr, tr, alpha="ACGT", shapes=("planted", "overlap", "random")):
    chunks, size, qo, ql, to, tl = [], 0, [], [], [], []
    for k in range(n_pairs):
        n = int(rng.randint(tr[0], tr[1] + 1)); m = int(rng.randint(qr[0], qr[1] + 1))
        t = helpers.rand_seq(rng, n, alpha)
        shape = shapes[k % len(shapes)]
        if shape == "planted" and n > 10:
            a = int(rng.randint(0, max(1, n - m))); q = helpers.mutate(rng, t[a:a + m], float(rng.choice([0.0, 0.07, 0.2]))) or "A"
        elif shape == "overlap":
            q = (helpers.rand_seq(rng, int(rng.randint(0, 30)), alpha) + helpers.mutate(rng, t[:max(1, n // 2)], 0.1)) or "C"
        else:
            q = helpers.rand_seq(rng, m, alpha)
        for s, offs, lens in ((q, qo, ql), (t, to, tl)):
            offs.append(size); lens.append(len(s)); chunks.append(s.encode()); size += len(s)
    return (np.frombuffer(b"".join(chunks), np.uint8), np.array(qo, np.int64), np.array(ql, np.int32), np.array(to, np.int64),
            np.array(tl, np.int32))


def _check(ctx, batch, **kw):
    from oracle import cport
    arena, qo, ql, to, tl = batch
    got = ctx.sg_affine_batch(arena, qo, ql, to, tl, **kw)
    only = ctx.sg_affine_batch(arena, qo, ql, to, tl, want_ops=False, **kw)
    for p in range(len(ql)):
        q = arena[qo[p]:qo[p] + ql[p]].tobytes(); t = arena[to[p]:to[p] + tl[p]].tobytes()
        want = cport.sg_affine(q, t, kw.get("match", 2), kw.get("mismatch", -2), kw.get("gap_open", 3), kw.get("gap_extend", 1))
        assert (int(got["score"][p]), int(got["end_query"][p]), int(got["end_ref"][p])) == (want["score"], want["end_query"], want["end_ref"]), p
        o = int(got["ops_off"][p])
        assert got["ops"][o:o + got["n_ops"][p]].tolist() == want["ops"], p
        assert (int(only["score"][p]), int(only["end_query"][p]), int(only["end_ref"][p])) == (want["score"], want["end_query"], want["end_ref"])
    return got


def test_small_and_ragged_pairs(ctx):
    rng = np.random.RandomState(1)
    _check(ctx, _batch(rng, 400, (1, 70), (1, 90)))
    _check(ctx, _batch(rng, 200, (60, 70), (1, 200)))                     # around one strip of 64 rows
    _check(ctx, _batch(rng, 120, (1, 200), (60, 70), alpha="ACGTNacgtn"))  # case-insensitive ACGT, N never matches


def test_read_sized_pairs(ctx):
    rng = np.random.RandomState(2)
    _check(ctx, _batch(rng, 60, (300, 1500), (300, 1600)))
    _check(ctx, _batch(rng, 4, (2500, 5000), (2500, 5200), shapes=("planted",)))


def test_other_scoring_and_empty_batch(ctx):
    rng = np.random.RandomState(3)
    _check(ctx, _batch(rng, 150, (5, 150), (5, 150)), match=1, mismatch=-3, gap_open=5, gap_extend=2)
    _check(ctx, _batch(rng, 100, (5, 150), (5, 150)), match=3, mismatch=-1, gap_open=1, gap_extend=1)
    r = ctx.sg_affine_batch(np.zeros(4, np.uint8), [], [], [], [])
    assert len(r["score"]) == 0


def test_bad_arguments(ctx):
    from ultra_amd import lib
    with pytest.raises(lib.UltraError):
        ctx.sg_affine_batch(np.zeros(8, np.uint8), [0], [4], [6], [9])       # target outside the arena
    with pytest.raises(lib.UltraError):
        ctx.sg_affine_batch(np.zeros(8, np.uint8), [0], [0], [2], [3])       # empty query


def test_get_exact_alignment_batch_mirrors_the_reference_wrapper(ctx):
    """align.get_exact_alignment: parasail route (gapped strings from the edit script, the score) and the edlib 'HW'
    route for very long sequences (with the reference's off-by-one slice of the located reference piece)."""
    from oracle import cport
    from ultra_amd import align
    rng = np.random.RandomState(9)
    jobs = []
    for k in range(40):
        ref = helpers.rand_seq(rng, int(rng.randint(200, 900)))
        a = int(rng.randint(0, 50))
        read = helpers.mutate(rng, ref[a:len(ref) - int(rng.randint(0, 50))], 0.08) or "ACGT"
        jobs.append((read, ref, len(ref)))
    long_ref = helpers.rand_seq(rng, 2400)
    jobs.append((helpers.mutate(rng, long_ref[300:1500], 0.05), long_ref, 20000))        # 1000 < len(read) < exons_len / 10
    got = align.get_exact_alignment_batch(jobs, ctx=ctx)
    for (read, ref, ln), (read_aln, ref_aln, score) in zip(jobs[:-1], got[:-1]):
        want = cport.sg_affine(read, ref)
        assert score == want["score"]
        assert read_aln.replace("-", "") == read and ref_aln.replace("-", "") == ref and len(read_aln) == len(ref_aln)
        shape = "".join("I" if b == "-" else "D" if a == "-" else "M" for a, b in zip(read_aln, ref_aln))
        assert shape == "".join("IDM"[{1: 0, 2: 1}.get(o, 2)] for o in want["ops"])
    read, ref, _ = jobs[-1]
    read_aln, ref_aln, score = got[-1]
    e = cport.hw_align(read, ref, -1)
    start, stop = e["locations"][0]
    assert len(read_aln) == len(ref_aln) and read_aln.startswith("-" * start) and ref_aln.startswith(ref[:start])
    assert read_aln.replace("-", "") == read
    assert score == sum(2 for a, b in zip(read_aln, ref_aln) if a == b) - 2 * e["editDistance"]
    assert align.get_exact_alignment(*jobs[3], ctx=ctx) == got[3]
