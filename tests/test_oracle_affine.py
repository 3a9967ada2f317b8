"""CPU: the affine-gap semi-global oracle (oracle/affine_port.c).  parasail is absent from
the image: the SCORE is pinned against an independent plain three-matrix DP written here, and every edit script is
checked to be a valid alignment of the two whole sequences that re-scores to exactly that score (free end gaps).
The choice among co-optimal paths / end cells is parasail's own and stays parity-unpinned (see the oracle's header)."""
import numpy as np

from oracle import cport
from tests import helpers

NEG = -10 ** 9


def plain_score(q, t, match=2, mismatch=-2, go=3, ge=1):
    m, n = len(q), len(t)
    H = [[0] * (n + 1) for _ in range(m + 1)]
    E = [[NEG] * (n + 1) for _ in range(m + 1)]
    F = [[NEG] * (n + 1) for _ in range(m + 1)]
    canon = lambda c: c.upper() if c.upper() in "ACGT" else None
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            E[i][j] = max(E[i][j - 1] - ge, H[i][j - 1] - go)
            F[i][j] = max(F[i - 1][j] - ge, H[i - 1][j] - go)
            a, b = canon(q[i - 1]), canon(t[j - 1])
            sub = 0 if a is None or b is None else match if a == b else mismatch     # wildcard row / column of matrix_create: 0
            H[i][j] = max(H[i - 1][j - 1] + sub, E[i][j], F[i][j])
    return max(max(H[m][1:]), max(H[i][n] for i in range(1, m + 1)))


def rescore(ops, q, t, match=2, mismatch=-2, go=3, ge=1):
    """Score of an edit script over the WHOLE of q and t with free end gaps; asserts it consumes both exactly."""
    i = j = 0
    cols = []
    for op in ops:
        if op in (0, 3):
            same = q[i].upper() == t[j].upper() and q[i].upper() in "ACGT"
            assert (op == 0) == same
            wild = q[i].upper() not in "ACGT" or t[j].upper() not in "ACGT"
            cols.append("M" if same else "W" if wild else "X"); i += 1; j += 1
        elif op == 1:
            cols.append("I"); i += 1
        else:
            cols.append("D"); j += 1
    assert i == len(q) and j == len(t)
    s = "".join(cols)
    # free end gaps: strip one leading run and one trailing run of gaps (either kind, the script has at most one kind per end)
    a = 0
    while a < len(s) and s[a] in "ID" and (a == 0 or s[a] == s[0]):
        a += 1
    b = len(s)
    while b > a and s[b - 1] in "ID" and (b == len(s) or s[b - 1] == s[-1]):
        b -= 1
    core = s[a:b]
    score, prev = 0, None
    for c in core:
        if c == "M":
            score += match
        elif c == "X":
            score += mismatch
        elif c == "W":
            pass
        else:
            score -= ge if prev == c else go
        prev = c
    return score


def test_score_equals_plain_dp_and_scripts_rescore():
    rng = np.random.RandomState(4)
    n_cases = 0
    for it in range(220):
        alpha = "ACGT" if it % 4 else "ACGTNacgtn"
        n = int(rng.randint(1, 90)); m = int(rng.randint(1, 90))
        t = helpers.rand_seq(rng, n, alpha)
        if it % 3 == 0 and n > 10:
            a = int(rng.randint(0, n - 5)); q = helpers.mutate(rng, t[a:a + m], float(rng.choice([0.0, 0.1, 0.3]))) or "A"
        elif it % 3 == 1:
            q = helpers.mutate(rng, t, 0.15)[int(rng.randint(0, 5)):] or "C"      # overlap: suffix/prefix shapes
        else:
            q = helpers.rand_seq(rng, m, alpha)
        r = cport.sg_affine(q, t)
        assert r["score"] == plain_score(q, t), (q, t)
        assert rescore(r["ops"], q, t) == r["score"], (q, t, r)
        assert r["end_query"] == len(q) - 1 or r["end_ref"] == len(t) - 1
        n_cases += 1
    assert n_cases == 220


def test_known_small_cases():
    r = cport.sg_affine("ACGT", "TTACGTTT")
    assert (r["score"], r["cigar"]) == (8, "2D4=2D")
    r = cport.sg_affine("ACGTACGT", "ACGACGT")       # one deleted base in the reference: gap open 3
    assert r["score"] == 2 * 7 - 3
    assert cport.sg_affine("AAAA", "CCCC")["score"] <= 0
    # worked by hand: N pairs with anything at 0.  ACNT in TTACGTTT: A=, C=, N/G 0, T= -> 6 (a mismatch would give 2 + 2 - 2 + 2 = 4)
    r = cport.sg_affine("ACNT", "TTACGTTT")
    assert (r["score"], r["cigar"]) == (6, "2D2=1X1=2D")
    assert cport.sg_affine("NNNN", "ACGT")["score"] == 0 and cport.sg_affine("acgt", "ACGT")["score"] == 8
    assert cport.sg_affine("ACGNACG", "ACGTACG")["score"] == 12 and cport.sg_affine("ACGAACG", "ACGTACG")["score"] == 10
