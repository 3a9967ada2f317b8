"""CPU: pins the Myers oracle (oracle/edlib_port.c).

edlib itself is absent from the reference tree and from this image, and the
reference holds no golden vector for the call, so:
  * distance / end columns / start columns are pinned mathematically: the 64-row block bit-vector
    port must equal the plain O(mn) DP of the same file on thousands of random pairs;
  * the traceback tie rule (up, then left, then diagonal) is pinned by the one known-answer vector
    of edlib's published Python README (align("ACTG", "CACTRT", mode="HW", task="path") ->
    distance 1, locations [(1, 3), (1, 4)], cigar "3=1I"), recalled from the public documentation -
    not from any file on disk - and otherwise documented as "parity unpinned" against real edlib.
"""
import numpy as np

from tests import helpers


def test_published_edlib_readme_example():
    from oracle import cport
    for method in ("bitvector", "bruteforce"):
        r = cport.hw_align("ACTG", "CACTRT", -1, method=method)
        assert r["editDistance"] == 1
        assert r["locations"] == [(1, 3), (1, 4)]
        assert r["cigar"] == "3=1I"
        assert r["alphabetLength"] == 5


def test_k_cutoff_and_exact_mode():
    from oracle import cport
    assert cport.hw_align("AAAAAAAA", "CCCCCCCCCC", 2)["editDistance"] == -1
    assert cport.hw_align("AAAAAAAA", "CCCCCCCCCC", 8)["editDistance"] == 8
    r = cport.hw_align("ACGTA", "TTACGTATTACGTAT", 0)
    assert r["editDistance"] == 0 and r["locations"] == [(2, 6), (9, 13)] and r["cigar"] == "5="
    # float k of the reference (0.4 * min(len)) is truncated: int(0.4 * n) == (2 * n) // 5
    for n in range(1, 5000):
        assert int(0.4 * n) == (2 * n) // 5


def test_bitvector_port_equals_plain_dp():
    from oracle import cport
    rng = np.random.RandomState(12)
    n_hit = 0
    for it in range(2500):
        alpha = ["ACGT", "AC", "ACGTN", "A"][it % 4]
        n = int(rng.randint(1, 260)); m = int(rng.randint(1, 200))
        t = helpers.rand_seq(rng, n, alpha)
        if rng.rand() < 0.6 and n > m:
            a = int(rng.randint(0, n - m + 1))
            q = helpers.mutate(rng, t[a:a + m], float(rng.choice([0, 0.05, 0.2, 0.4]))) or "A"
        else:
            q = helpers.rand_seq(rng, m, alpha)
        k = [int(0.4 * min(len(q), n)), 0, -1, len(q), 2][it % 5]
        a = cport.hw_align(q, t, k, method="bitvector")
        b = cport.hw_align(q, t, k, method="bruteforce")
        assert a == b, (q, t, k)
        n_hit += a["editDistance"] >= 0
    assert n_hit > 1000


def test_multiword_queries_equal_plain_dp():
    from oracle import cport
    rng = np.random.RandomState(13)
    for it in range(60):
        m = int(rng.choice([63, 64, 65, 127, 128, 129, 191, 192, 193, 300, 520]))
        n = int(rng.randint(m // 2, 3 * m))
        t = helpers.rand_seq(rng, n)
        q = helpers.mutate(rng, (t + t)[n // 3:n // 3 + m], 0.1) if it % 3 else helpers.rand_seq(rng, m)
        k = int(0.4 * min(len(q), n))
        assert cport.hw_align(q, t, k) == cport.hw_align(q, t, k, method="bruteforce")


def test_batch_driver_matches_single_calls():
    from oracle import cport
    rng = np.random.RandomState(14)
    arena, q_off, q_len, t_off, t_len, k = helpers.make_pairs(rng, 300, (5, 150), (30, 400), kmode="mixed")
    loc_off = np.arange(len(k) + 1, dtype=np.int64) * 400
    ed, nl, st, en, fi, ovf = cport.hw_batch(arena, q_off, q_len, t_off, t_len, k, loc_off)
    assert ovf == 0
    for p in range(len(k)):
        q = bytes(arena[q_off[p]:q_off[p] + q_len[p]]); t = bytes(arena[t_off[p]:t_off[p] + t_len[p]])
        r = cport.hw_align(q, t, int(k[p]))
        assert r["editDistance"] == ed[p]
        o = int(loc_off[p])
        assert r["locations"] == list(zip(st[o:o + nl[p]].tolist(), en[o:o + nl[p]].tolist()))
        assert r["insertions"] == fi[p]


def _all_optimal_ins_counts(q, t):
    """Every insertion count an optimal global alignment of q with t can have (plain DP + walk over tight moves)."""
    m, n = len(q), len(t)
    D = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(m + 1):
        for j in range(n + 1):
            if i == 0 or j == 0:
                D[i][j] = i + j
            else:
                D[i][j] = min(D[i - 1][j - 1] + (q[i - 1] != t[j - 1]), D[i - 1][j] + 1, D[i][j - 1] + 1)
    memo = {}

    def counts(i, j):
        if (i, j) not in memo:
            if i == 0 and j == 0:
                memo[(i, j)] = {0}
            else:
                s = set()
                if i > 0 and D[i - 1][j] + 1 == D[i][j]:
                    s |= {c + 1 for c in counts(i - 1, j)}
                if j > 0 and D[i][j - 1] + 1 == D[i][j]:
                    s |= counts(i, j - 1)
                if i > 0 and j > 0 and D[i - 1][j - 1] + (q[i - 1] != t[j - 1]) == D[i][j]:
                    s |= counts(i - 1, j - 1)
                memo[(i, j)] = s
        return memo[(i, j)]
    return D[m][n], counts(m, n)


def test_insertion_range_over_all_optimal_paths():
    """oracle_ins_range brackets every optimal path's insertion count (the exposure of the unpinned traceback rule)."""
    from oracle import cport
    assert cport.ins_range("ACTG", "ACT", 1) == (1, 1)
    assert cport.ins_range("AB", "BA", 2) == (0, 1)        # two substitutions, or insert / match / delete
    rng = np.random.RandomState(15)
    wide = 0
    for it in range(1500):
        alpha = ["ACGT", "AC", "A"][it % 3]
        n = int(rng.randint(1, 14)); m = int(rng.randint(1, 14))
        t = helpers.rand_seq(rng, n, alpha)
        q = helpers.mutate(rng, t, 0.3) or "A" if it % 2 else helpers.rand_seq(rng, m, alpha)
        ed, cs = _all_optimal_ins_counts(q, t)
        lo, hi = cport.ins_range(q, t, ed)
        assert (lo, hi) == (min(cs), max(cs)), (q, t)
        wide += hi > lo
    assert wide > 100
    # the oracle's own path (rule: up, then left, then diagonal) lies inside the range, on real-sized hits
    for it in range(200):
        t = helpers.rand_seq(rng, int(rng.randint(200, 900)))
        m = int(rng.randint(20, 180)); a = int(rng.randint(0, len(t) - m))
        q = helpers.mutate(rng, t[a:a + m], float(rng.choice([0.03, 0.1, 0.25]))) or "A"
        r = cport.hw_align(q, t, int(0.4 * min(len(q), len(t))))
        if r["editDistance"] < 0:
            continue
        s, e = r["locations"][0]
        lo, hi = cport.ins_range(q, t[s:e + 1], r["editDistance"])
        assert lo <= r["insertions"] <= hi


def test_insertion_count_is_arithmetic_when_at_most_one_edit_is_free():
    """What the banded traceback on the device relies on (csrc/myers.hip, tb_by_arithmetic): a global alignment of m query
    bases with L target bases at distance ed has I - D = m - L and X + I + D = ed, so with ed - |m - L| <= 1 every optimal
    path has I = max(m - L, 0) - whatever the tie rule.  Checked on the oracle's own traceback (cigar insertions of the
    first location) over random pairs, all three kinds of edit and long gaps included; both kinds of hit must occur."""
    from oracle import cport
    rng = np.random.RandomState(31)
    free = walked = 0
    for mix, block in (((0.5, 0.3, 0.2), 0), ((1.0, 0.0, 0.0), 0), ((0.0, 0.0, 1.0), 0), ((0.45, 0.1, 0.45), 40)):
        arena, q_off, q_len, t_off, t_len, k = helpers.make_pairs(rng, 600, (5, 160), (60, 500), planted=0.9,
                                                                  rates=(0.0, 0.01, 0.03, 0.08, 0.2), mix=mix, block=block)
        for p in range(len(k)):
            q = bytes(arena[q_off[p]:q_off[p] + q_len[p]]); t = bytes(arena[t_off[p]:t_off[p] + t_len[p]])
            r = cport.hw_align(q, t, int(k[p]))
            ed = r["editDistance"]
            if ed < 0:
                continue
            s, e = r["locations"][0]
            m, L = len(q), e - s + 1
            ins, num = 0, ""
            for ch in r["cigar"]:
                if ch.isdigit():
                    num += ch
                else:
                    if ch == "I":
                        ins += int(num)
                    num = ""
            if ed - abs(m - L) <= 1:
                assert ins == max(m - L, 0), (p, ed, m, L, r["cigar"])
                free += 1
            else:
                walked += 1
    assert free > 300 and walked > 300
