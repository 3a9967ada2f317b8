"""GPU parity: libultra_hip Myers batch (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _check(ctx, pairs, want_ops=False):
    from oracle import cport
    arena, q_off, q_len, t_off, t_len, k = pairs
    got = ctx.myers_hw_batch(arena, q_off, q_len, t_off, t_len, k, want_ops=want_ops)
    cap = np.maximum(got["n_loc"].astype(np.int64), 1) + 1
    loc_off = np.zeros(len(k) + 1, np.int64); np.cumsum(cap, out=loc_off[1:])
    ed, nl, st, en, fi, ovf = cport.hw_batch(arena, q_off, q_len, t_off, t_len, k, loc_off)
    assert ovf == 0
    assert np.array_equal(got["ed"], ed)
    assert np.array_equal(got["n_loc"], nl)
    assert np.array_equal(got["first_ins"], fi)
    for p in range(len(k)):
        a, b = int(got["loc_off"][p]), int(got["loc_off"][p + 1])
        o = int(loc_off[p])
        assert np.array_equal(got["loc_end"][a:b], en[o:o + nl[p]]), p
        assert np.array_equal(got["loc_start"][a:b], st[o:o + nl[p]]), p
    if want_ops:
        for p in range(len(k)):
            if ed[p] < 0:
                continue
            q = bytes(arena[q_off[p]:q_off[p] + q_len[p]]); t = bytes(arena[t_off[p]:t_off[p] + t_len[p]])
            ref = cport.hw_align(q, t, int(k[p]))
            n = int(got["n_ops"][p])
            ops = got["ops"][int(got["ops_off"][p]):int(got["ops_off"][p]) + n][::-1]
            assert cport.ops_to_cigar(ops.tolist()) == ref["cigar"], p
    return got


@pytest.mark.parametrize("qr,tr,n", [((1, 64), (20, 400), 3000), ((60, 140), (100, 1200), 3000),
                                      ((120, 520), (300, 1500), 1500)])
def test_lane_kernels_match_oracle(ctx, qr, tr, n):
    rng = np.random.RandomState(hash((qr, tr)) % (2 ** 31))
    _check(ctx, helpers.make_pairs(rng, n, qr, tr, kmode="mixed"))


def test_group_kernels_match_oracle(ctx):
    rng = np.random.RandomState(77)
    _check(ctx, helpers.make_pairs(rng, 120, (513, 2100), (400, 2600), kmode="ref"))
    _check(ctx, helpers.make_pairs(rng, 24, (2100, 9000), (1500, 9000), kmode="ref", rates=(0.02, 0.1)))


def test_small_exact_and_many_locations(ctx):
    # 5-8 bp exact segments (k = 0) and low-complexity targets: many end positions, slot overflow path
    rng = np.random.RandomState(5)
    _check(ctx, helpers.make_pairs(rng, 2000, (5, 8), (50, 1000), kmode="zero", alpha="AC"))
    _check(ctx, helpers.make_pairs(rng, 500, (5, 30), (200, 1000), kmode="ref", alpha="A"))
    _check(ctx, helpers.make_pairs(rng, 500, (9, 40), (200, 800), kmode="ref", alpha="AC"))


def test_alphabet_is_byte_exact(ctx):
    rng = np.random.RandomState(6)
    _check(ctx, helpers.make_pairs(rng, 800, (9, 200), (100, 600), kmode="ref", alpha="ACGTNacgtn"))


IUPAC33 = "ACGTUNRYKMSWBVDH" + "acgtunrykmswbvdh" + "X"     # every byte the reference's reverse_complement knows


@pytest.mark.parametrize("alpha", ["ACGTNacgtnRYKMSWBVDH", IUPAC33])
def test_soft_masked_iupac_alphabets(ctx, alpha):
    """More than 16 distinct bytes in a batch (soft-masked sequence + IUPAC codes): compared byte-exact and
    case-sensitive like everything else (the reference never upper-cases), through every size class, the banded
    traceback (up to 31 symbols) and the full-height one (beyond)."""
    rng = np.random.RandomState(len(alpha))
    got = _check(ctx, helpers.make_pairs(rng, 900, (5, 300), (100, 900), kmode="mixed", planted=0.8, alpha=alpha))
    assert (got["ed"] >= 0).sum() > 300
    _check(ctx, helpers.make_pairs(rng, 60, (300, 2600), (400, 3000), kmode="ref", planted=0.9, alpha=alpha, rates=(0.02, 0.1)))
    _check(ctx, helpers.make_pairs(rng, 6, (2600, 9000), (3000, 9000), kmode="ref", planted=0.9, alpha=alpha, rates=(0.02, 0.1)))
    _check(ctx, helpers.make_pairs(rng, 200, (9, 200), (100, 500), kmode="ref", planted=0.8, alpha=alpha), want_ops=True)


def test_edit_scripts(ctx):
    rng = np.random.RandomState(8)
    _check(ctx, helpers.make_pairs(rng, 600, (9, 300), (100, 700), kmode="ref", planted=0.8), want_ops=True)
    _check(ctx, helpers.make_pairs(rng, 30, (600, 1500), (800, 2000), kmode="ref", planted=0.9), want_ops=True)


def test_query_longer_than_target_and_contained_reads(ctx):
    rng = np.random.RandomState(9)
    # read-in-segment shape: query (read) about as long as or longer than the target (segment)
    _check(ctx, helpers.make_pairs(rng, 300, (200, 600), (150, 700), kmode="ref", planted=0.0))


def test_empty_batch_and_bad_arguments(ctx):
    from ultra_amd import lib
    r = ctx.myers_hw_batch(np.zeros(4, np.uint8), [], [], [], [], [])
    assert len(r["ed"]) == 0 and r["loc_off"][0] == 0
    with pytest.raises(lib.UltraError):
        ctx.myers_hw_batch(np.zeros(4, np.uint8), [0], [3], [2], [9], [1])      # target outside the arena
    with pytest.raises(lib.UltraError):
        ctx.myers_hw_batch(np.zeros(4, np.uint8), [0], [0], [0], [4], [1])      # empty query
    with pytest.raises(lib.UltraError):
        ctx.myers_hw_batch(np.arange(64, dtype=np.uint8), [0], [10], [10], [50], [5])  # > 36 distinct symbols


def test_round_trip_property_at_scale(ctx):
    """Size-independent property at a bench-like size: every reported (start, end) re-aligns GLOBALLY to exactly ed
    (plain DP, no oracle involved), no earlier start does, and no end column outside the list reaches ed."""
    rng = np.random.RandomState(10)
    pairs = helpers.make_pairs(rng, 20000, (9, 300), (800, 1200), kmode="ref")
    arena, q_off, q_len, t_off, t_len, k = pairs
    got = ctx.myers_hw_batch(arena, q_off, q_len, t_off, t_len, k)
    hits = np.nonzero(got["ed"] >= 0)[0]
    assert len(hits) > 5000
    for p in hits[:: max(1, len(hits) // 400)]:
        q = arena[q_off[p]:q_off[p] + q_len[p]]; t = arena[t_off[p]:t_off[p] + t_len[p]]
        ed = int(got["ed"][p])
        a, b = int(got["loc_off"][p]), int(got["loc_off"][p + 1])
        assert b > a
        bottom = helpers.semiglobal_bottom_row(q, t)
        assert int(bottom.min()) == ed
        assert np.array_equal(np.nonzero(bottom == ed)[0], got["loc_end"][a:b])
        for s, e in zip(got["loc_start"][a:b], got["loc_end"][a:b]):
            assert helpers.nw_distance(q, t[s:e + 1]) == ed, (p, s, e)
            for s2 in range(max(0, s - 3), s):      # start = SMALLEST start that attains ed
                assert helpers.nw_distance(q, t[s2:e + 1]) > ed, (p, s2, e)


def test_banded_and_full_height_traceback_agree(ctx):
    """The banded traceback (default) and the full-height one (cfg banded_traceback=0, also the
    fallback for distances > 374) give the same insertion counts; both equal the oracle."""
    from ultra_amd import lib
    rng = np.random.RandomState(11)
    full = lib.Context(0, cfg={"banded_traceback": 0})
    try:
        for qr, tr, n, rates in [((9, 300), (100, 900), 1500, (0.0, 0.05, 0.2, 0.35)),
                                 ((300, 1400), (400, 2000), 200, (0.02, 0.1, 0.2)),
                                 ((1300, 1900), (1500, 2600), 60, (0.3, 0.36)),    # the 8-, 12- and 16-word bands (lane groups)
                                 ((3600, 4400), (3800, 5200), 20, (0.36, 0.39))]:  # distances beyond every band class (> 1014)
            pairs = helpers.make_pairs(rng, n, qr, tr, kmode="ref", planted=0.9, rates=rates)
            a = _check(ctx, pairs)
            b = _check(full, pairs)
            assert np.array_equal(a["first_ins"], b["first_ins"])
            # both kinds of hit are in the batch: insertion count by arithmetic (ed - |m - L| <= 1, no traceback in the
            # banded mode) and by traceback
            hit = np.flatnonzero(a["ed"] >= 0)
            first = a["loc_off"][hit]
            span = (a["loc_end"][first] - a["loc_start"][first] + 1).astype(np.int64)
            free = a["ed"][hit] - np.abs(np.asarray(pairs[2])[hit].astype(np.int64) - span) <= 1
            assert free.any() or qr[0] >= 300
            assert (~free).any()
            assert (a["ed"] > 374).any() or qr[0] < 1300
            assert (a["ed"] > 1014).any() or qr[0] < 3600
    finally:
        full.close()


def test_banded_traceback_wide_alphabet(ctx):
    rng = np.random.RandomState(12)
    _check(ctx, helpers.make_pairs(rng, 600, (9, 400), (100, 900), kmode="ref", planted=0.9, alpha="ACGTNRYKMSWacg"))


@pytest.mark.parametrize("mix", [(1.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.45, 0.1, 0.45), (0.5, 0.3, 0.2)])
def test_banded_traceback_skewed_paths(ctx, mix):
    """The band is shifted per pair towards the diagonals its path can reach (end diagonal m - L): deletion-only,
    insertion-only and long contiguous gaps put the path far from the main diagonal, and the rates make the
    distances straddle the band-class limits (22, 54, 118, 246, 374; 502, 758, 1014 are covered by
    test_banded_and_full_height_traceback_agree and test_wide_lane_group_bands)."""
    rng = np.random.RandomState(13 + int(mix[0] * 10) + int(mix[2] * 100))
    for qr, tr, n, rates, block in [((9, 260), (100, 900), 1200, (0.05, 0.2, 0.3, 0.38), 0),
                                    ((100, 400), (300, 1000), 500, (0.1, 0.2, 0.3), 60),
                                    ((250, 800), (500, 1600), 250, (0.08, 0.16, 0.3), 120),
                                    ((600, 1500), (900, 2400), 80, (0.08, 0.17, 0.33), 200)]:
        got = _check(ctx, helpers.make_pairs(rng, n, qr, tr, kmode="ref", planted=0.9, rates=rates, mix=mix, block=block))
        assert (got["ed"] > 54).any()
        if block == 0:
            assert ((got["ed"] >= 0) & (got["ed"] <= 22)).any() and ((got["ed"] > 22) & (got["ed"] <= 54)).any()


def test_banded_and_full_height_start_pass_agree(ctx):
    """Start positions from the full-height reverse pass (default) and from the banded one (cfg banded_starts=1: lane
    per location for distances <= 59, dense full-height lists beyond) are the same and equal the oracle: many
    locations per pair (low-complexity sequences), locations whose window runs into the target's first column,
    distances on either side of every band-class limit (5, 11, 27, 59), queries in every size class."""
    from ultra_amd import lib
    rng = np.random.RandomState(21)
    full = lib.Context(0, cfg={"banded_starts": 1})
    try:
        cases = [((5, 64), (20, 300), 2500, (0.0, 0.1, 0.25, 0.38), "ACGT"),
                 ((5, 40), (40, 400), 1500, (0.0, 0.2), "AC"),                       # dozens of tied end columns
                 ((40, 300), (60, 500), 2000, (0.02, 0.08, 0.2, 0.35), "ACGT"),
                 ((200, 700), (150, 900), 700, (0.05, 0.1, 0.2, 0.3), "ACGT"),       # targets shorter than queries too
                 ((600, 1500), (900, 2400), 120, (0.03, 0.08, 0.13, 0.3), "ACGT"),
                 ((1500, 2600), (2000, 3500), 40, (0.02, 0.06, 0.12), "ACGTN")]
        seen = set()
        for qr, tr, n, rates, alpha in cases:
            pairs = helpers.make_pairs(rng, n, qr, tr, kmode="ref", planted=0.9, rates=rates, alpha=alpha)
            a = _check(ctx, pairs)
            b = _check(full, pairs)
            assert np.array_equal(a["loc_start"], b["loc_start"]) and np.array_equal(a["loc_end"], b["loc_end"])
            ed = a["ed"][a["ed"] >= 0]
            for lim in (5, 11, 27, 59, 123, 187):
                if (ed <= lim).any() and (ed > lim).any():
                    seen.add(lim)
        assert seen == {5, 11, 27, 59, 123, 187}
        assert (a["n_loc"] >= 1).any()
    finally:
        full.close()


@pytest.mark.parametrize("mix", [(0.5, 0.3, 0.2), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)])
def test_wide_lane_group_bands(ctx, mix):
    """The band classes of 4 to 16 words run with 2 to 8 lanes per hit (myers_bandg): distances on either side of
    246, 374, 502, 758 and 1014, balanced / deletion-only / insertion-only edits and long contiguous gaps, several
    hits per wavefront and wavefronts that are not full."""
    rng = np.random.RandomState(31 + int(mix[0] * 10) + int(mix[2] * 100))
    seen = set()
    for qr, tr, n, rates, block in [((700, 1400), (900, 2600), 150, (0.2, 0.28, 0.36), 0),
                                    ((1500, 2600), (1800, 3600), 70, (0.12, 0.2, 0.3, 0.38), 300),
                                    ((2400, 3000), (2600, 4000), 23, (0.2, 0.3, 0.36), 0)]:
        got = _check(ctx, helpers.make_pairs(rng, n, qr, tr, kmode="ref", planted=0.95, rates=rates, mix=mix, block=block))
        ed = got["ed"][got["ed"] >= 0]
        for lim in (246, 374, 502, 758, 1014):
            if (ed <= lim).any() and (ed > lim).any():
                seen.add(lim)
    assert {374, 502} <= seen and (758 in seen or mix[0] == 1.0)   # deletions alone stay below 758 at these lengths


def _pack(pairs):
    """[(query, target, k)] of str -> the arrays of ultra_myers_hw_batch."""
    chunks, size, q_off, q_len, t_off, t_len, ks = [], 0, [], [], [], [], []
    for q, t, k in pairs:
        for s, offs, lens in ((q, q_off, q_len), (t, t_off, t_len)):
            offs.append(size); lens.append(len(s)); chunks.append(s.encode()); size += len(s)
        ks.append(k)
    return (np.frombuffer(b"".join(chunks), dtype=np.uint8), np.array(q_off, np.int64), np.array(q_len, np.int32),
            np.array(t_off, np.int64), np.array(t_len, np.int32), np.array(ks, np.int32))


def test_length_prefilter_on_both_sides_of_the_bound(ctx):
    """A query more than k bases longer than its target cannot be a hit (distance >= m - n): the library answers those
    pairs when the batch is staged and no kernel sees them (ultra_myers_cell_counts).  Pairs on either side of
    m - n = k, in every size class, with queries that DO reach distance m - n (the target with bases spliced in), so
    that the pair next to the bound is a hit and the one beyond is not: results equal the oracle's either way."""
    from oracle import cport
    rng = np.random.RandomState(41)
    pairs, expect_skipped = [], 0
    for n in (12, 60, 130, 250, 400, 700, 1500, 2500, 5000):
        for extra in (1, 7, n // 3, n // 2):
            t = helpers.rand_seq(rng, n)
            q = list(t)
            for _ in range(extra):
                q.insert(int(rng.randint(0, len(q) + 1)), "ACGT"[rng.randint(0, 4)])
            q = "".join(q)
            for k in (extra - 2, extra - 1, extra, extra + 1, int(0.4 * n), -1, len(q) + 5):
                pairs.append((q, t, k))
                ke = len(q) if (k < 0 or k > len(q)) else k
                expect_skipped += (len(q) - n) > ke
            pairs.append((helpers.rand_seq(rng, n + extra), t, int(0.4 * n)))        # the reference's k, unrelated query
            expect_skipped += extra > int(0.4 * n)
    packed = _pack(pairs)
    got = _check(ctx, packed)
    assert (got["ed"] >= 0).sum() > 100 and (got["ed"] < 0).sum() > 100
    ctx.myers_stage(*packed)
    computed, skipped, n_skipped = ctx.myers_cell_counts()
    assert n_skipped == expect_skipped and 0 < n_skipped < len(pairs)
    arena, q_off, q_len, t_off, t_len, k = packed
    ke = np.where((k < 0) | (k > q_len), q_len, k)
    sk = (q_len - t_len) > ke
    cells = q_len.astype(np.float64) * t_len
    assert computed == cells[~sk].sum() and skipped == cells[sk].sum()
    ctx.myers_run()
    again = ctx.myers_fetch()
    assert np.array_equal(again["ed"], got["ed"]) and np.array_equal(again["loc_start"], got["loc_start"])
    assert (again["ed"][sk] == -1).all() and (again["n_loc"][sk] == 0).all()
    # and the oracle, which computes every cell, agrees that none of them is a hit
    lo = np.arange(len(k) + 1, dtype=np.int64) * 64
    assert (cport.hw_batch(arena, q_off, q_len, t_off, t_len, k, lo)[0][sk] == -1).all()


def test_queries_beyond_32768_rows(ctx):
    """Queries longer than the widest wavefront-per-pair class (32768 rows) run in strips of 32768 rows (one
    wavefront, strip after strip, the row between two strips handed over through HBM): a 40 kb read contained in a
    50 kb exon (classify_read_with_mams.py:374-377), the 50 kb exon as the query against the read (:312; m - n = 10 kb
    is below k = 16 kb, so the pair is not answered by length), a 70 kb query (three strips), low and high error rates
    (banded and full-height traceback), in an otherwise ordinary batch."""
    rng = np.random.RandomState(43)
    seg = helpers.rand_seq(rng, 50000)
    read = helpers.mutate(rng, seg[6000:46000], 0.07)
    clean = helpers.mutate(rng, seg[1000:41000], 0.01)
    big_t = helpers.rand_seq(rng, 72000)
    big_q = helpers.mutate(rng, big_t[500:70500], 0.008)
    pairs = [(read, seg, int(0.4 * len(read))),       # contained read, ~2800 edits: full-height traceback in strips
             (seg, read, int(0.4 * len(read))),       # the exon as the query: a hit through ~11 k inserted bases
             (clean, seg, int(0.4 * len(clean))),     # ~400 edits: banded traceback
             (big_q, big_t, int(0.4 * len(big_q))),   # three strips
             (helpers.rand_seq(rng, 33000), helpers.rand_seq(rng, 34000), -1),  # unrelated, unbounded k: hit with a huge distance
             (seg[:32769], seg[100:33100], 4000)]     # one row into the second strip
    normal = helpers.make_pairs(rng, 300, (9, 2600), (200, 3000), kmode="ref", planted=0.7)
    na, nqo, nql, nto, ntl, nk = normal
    for p in range(len(nk)):
        pairs.append((bytes(na[nqo[p]:nqo[p] + nql[p]]).decode(), bytes(na[nto[p]:nto[p] + ntl[p]]).decode(), int(nk[p])))
    order = rng.permutation(len(pairs))
    got = _check(ctx, _pack([pairs[i] for i in order]))
    ed = got["ed"][np.argsort(order)]
    assert ed[0] > 1014 and ed[1] > 10000 and 0 < ed[2] <= 1014 and 0 < ed[3] <= 1014 and ed[4] > 10000 and ed[5] >= 0


def test_more_than_2_pow_24_hits_in_one_batch(ctx):
    """A batch with more than 2^24 hits (a million cDNA reads make 19 million): the hit offsets come out of one packed scan
    whose hit field was 24 bits wide at first - tracebacks past it were lost without an error.  One tile of short pairs,
    checked against the oracle, repeated over the same arena bytes: every repetition must equal the first."""
    rng = np.random.RandomState(2024)
    tile = helpers.make_pairs(rng, 4096, (12, 28), (24, 60), kmode="ref")
    first = _check(ctx, tile)
    arena, q_off, q_len, t_off, t_len, k = tile
    keep = np.flatnonzero(first["ed"] >= 0)
    keep = np.sort(np.concatenate([keep, np.flatnonzero(first["ed"] < 0)[:64]]))
    n_hit = int((first["ed"][keep] >= 0).sum())
    assert n_hit > 2000 and first["first_ins"][keep].max() > 0
    reps = ((1 << 24) + 8192) // n_hit + 1
    cols = [np.tile(np.asarray(c)[keep], reps) for c in (q_off, q_len, t_off, t_len, k)]
    got = ctx.myers_hw_batch(arena, *cols)
    T = len(keep)
    assert int((got["ed"] >= 0).sum()) == reps * n_hit > (1 << 24)
    for f in ("ed", "n_loc", "first_ins"):
        assert np.array_equal(got[f].reshape(reps, T), np.broadcast_to(first[f][keep], (reps, T))), f
    per = int(got["loc_off"][T])
    assert int(got["loc_off"][-1]) == reps * per
    for f in ("loc_start", "loc_end"):
        rows = got[f][:reps * per].reshape(reps, per)
        assert np.array_equal(rows, np.broadcast_to(rows[0], (reps, per))), f
    a = np.concatenate([np.arange(first["loc_off"][p], first["loc_off"][p + 1]) for p in keep]).astype(np.int64)
    assert np.array_equal(got["loc_start"][:per], first["loc_start"][a]) and np.array_equal(got["loc_end"][:per], first["loc_end"][a])
