"""Shared generators for the parity tests (seeded, no reference code involved)."""
import numpy as np


def mutate(rng, s, rate, mix=(0.5, 0.3, 0.2)):
    """mix = shares of (deleted, substituted, inserted-after) positions among the erroneous ones."""
    out = []
    for ch in s:
        r = rng.rand()
        if r < rate * mix[0]:
            continue
        if r < rate * (mix[0] + mix[1]):
            out.append("ACGT"[rng.randint(0, 4)])
        elif r < rate:
            out.append(ch)
            out.append("ACGT"[rng.randint(0, 4)])
        else:
            out.append(ch)
    return "".join(out)


def rand_seq(rng, n, alpha="ACGT"):
    return "".join(alpha[i] for i in rng.randint(0, len(alpha), size=n))


def make_pairs(rng, n_pairs, qlen_range, tlen_range, planted=0.5, rates=(0.0, 0.02, 0.07, 0.15, 0.3),
               alpha="ACGT", kmode="ref", mix=(0.5, 0.3, 0.2), block=0):
    """Returns (arena uint8, q_off, q_len, t_off, t_len, k) with half the queries planted in the target.
    block > 0: a planted query also gets one contiguous gap of up to `block` bases (cut out or spliced in)."""
    chunks, size = [], 0
    q_off, q_len, t_off, t_len, ks = [], [], [], [], []
    for _ in range(n_pairs):
        n = int(rng.randint(tlen_range[0], tlen_range[1] + 1))
        m = int(rng.randint(qlen_range[0], qlen_range[1] + 1))
        t = rand_seq(rng, n, alpha)
        if rng.rand() < planted and n > m:
            a = int(rng.randint(0, n - m + 1))
            q = mutate(rng, t[a:a + m], rates[int(rng.randint(0, len(rates)))], mix) or "A"
            if block and len(q) > 4:
                g = int(rng.randint(1, block + 1)); at = int(rng.randint(1, len(q) - 1))
                q = (q[:at] + q[at + g:]) if rng.rand() < 0.5 else (q[:at] + rand_seq(rng, g, alpha) + q[at:])
                q = q or "A"
        else:
            q = rand_seq(rng, m, alpha)
        if kmode == "ref":
            k = int(0.4 * min(len(q), n))
        elif kmode == "zero":
            k = 0
        elif kmode == "mixed":
            k = [int(0.4 * min(len(q), n)), 0, len(q), -1, 3][int(rng.randint(0, 5))]
        else:
            k = int(kmode)
        for s, offs, lens in ((q, q_off, q_len), (t, t_off, t_len)):
            offs.append(size); lens.append(len(s))
            chunks.append(s.encode()); size += len(s)
        ks.append(k)
    arena = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    return arena, np.array(q_off, np.int64), np.array(q_len, np.int32), np.array(t_off, np.int64), \
        np.array(t_len, np.int32), np.array(ks, np.int32)


def _dp_rows(q, t, free_start):
    """Unit-cost DP of q (rows) against t (columns), row by row with numpy; returns the last row D[m][0..n].
    free_start: D[0][j] = 0 (semi-global) instead of j (global)."""
    q = np.frombuffer(q.encode(), np.uint8) if isinstance(q, str) else np.asarray(q, np.uint8)
    t = np.frombuffer(t.encode(), np.uint8) if isinstance(t, str) else np.asarray(t, np.uint8)
    n = len(t)
    idx = np.arange(n + 1, dtype=np.int64)
    prev = np.zeros(n + 1, np.int64) if free_start else idx.copy()
    for i in range(1, len(q) + 1):
        cur = np.empty(n + 1, np.int64)
        cur[0] = i
        cur[1:] = np.minimum(prev[:-1] + (t != q[i - 1]), prev[1:] + 1)
        # horizontal moves: cur[j] = min over j' <= j of cur[j'] + (j - j')
        cur = np.minimum.accumulate(cur - idx) + idx
        prev = cur
    return prev


def nw_distance(q, t):
    """Global edit distance (plain DP, independent of the oracle and of the library)."""
    return int(_dp_rows(q, t, False)[-1])


def semiglobal_bottom_row(q, t):
    """D[m][j] for end columns j = 0 .. n-1 of the infix search (0-based inclusive end), plain DP."""
    return _dp_rows(q, t, True)[1:]


def make_mams(rng, n, readlen, ties):
    """Random MAM instance sorted by y: arrays x, y, c, d, val."""
    rows = []
    for _ in range(n):
        if ties:
            L = int(rng.choice([10, 20, 30]))
            c = int(rng.choice(np.arange(0, max(1, readlen - L), 5)))
            x = int(rng.choice(np.arange(0, 5000, 50)))
            y = x + L
            val = float(L) * float(rng.choice([1.0, 0.9]))
        else:
            L = int(rng.randint(5, 61))
            c = int(rng.randint(0, max(1, readlen - L)))
            x = int(rng.randint(0, 5001))
            y = x + L + int(rng.randint(0, 4))
            val = float(L) if rng.rand() < 0.3 else L * float(rng.choice([0.7, 0.8, 0.9, 1.0, rng.rand()]))
        rows.append((x, y, c, c + L - 1, val))
    rows.sort(key=lambda r: r[1])
    a = np.array([r[:4] for r in rows], np.int32).reshape(-1, 4)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy(), a[:, 3].copy(), np.array([r[4] for r in rows], np.float64)


# ---- golden fixtures (tests/golden/*.json) ---------------------------------------------------------
def load_golden_world(rec):
    """Rebuilds the five index dictionaries of a classify_golden.json world record, in the layout the
    reference's hot path reads (keys = array('L',[chr,start,stop]).tobytes(), values = flat array('L'))."""
    from array import array

    def key(t):
        return array('L', [int(v) for v in t]).tobytes()
    seg = {key(r[:3]): r[3] for r in rec["segments"]}
    flank = {key(r[:3]): r[3] for r in rec["flanks"]}
    p2s = {key(k): array('L', flat) for k, flat in rec["parts_to_segments"]}
    s2g = {key(k): set(g) for k, g in rec["segment_to_gene"]}
    g2s = {g: array('L', flat) for g, flat in rec["gene_to_small_segments"]}
    return seg, flank, p2s, s2g, g2s


def golden_solution(rows):
    from ultra_amd.synth import Mem
    return [Mem(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), r[6]) for r in rows]


def mam_rows(sol):
    """Canonical comparable form of a MAM tuple list (val as IEEE-754 hex)."""
    return [[int(m.x), int(m.y), int(m.c), int(m.d), float(m.val).hex(), int(m.j), int(m.min_segment_length),
             str(m.mam_id), int(m.ref_chr_id)] for m in sol]


# ---- transcript of downstream calls (tests/golden/make_align_single_golden.py and its replay) ----
def digest(obj):
    """Stable digest of call arguments (floats by their bits, tuples as lists, sets sorted)."""
    import hashlib
    import json

    def norm(v):
        if isinstance(v, float):
            return v.hex()
        if isinstance(v, (list, tuple)):
            return [norm(x) for x in v]
        if isinstance(v, (set, frozenset)):
            return sorted(norm(x) for x in v)
        if isinstance(v, dict):
            return sorted([norm(k), norm(x)] for k, x in v.items())
        if isinstance(v, np.integer):
            return int(v)
        return v
    return hashlib.sha1(json.dumps(norm(obj), sort_keys=True).encode()).hexdigest()[:20]


def jsonable(v):
    if isinstance(v, float):
        return {"f": v.hex()}
    if isinstance(v, tuple):
        return {"t": [jsonable(x) for x in v]}
    if isinstance(v, list):
        return [jsonable(x) for x in v]
    if isinstance(v, (set, frozenset)):
        return {"s": sorted(jsonable(x) for x in v)}
    if isinstance(v, np.integer):
        return int(v)
    return v


def unjsonable(v):
    if isinstance(v, dict):
        if "f" in v:
            return float.fromhex(v["f"])
        if "t" in v:
            return tuple(unjsonable(x) for x in v["t"])
        return set(unjsonable(x) for x in v["s"])
    if isinstance(v, list):
        return [unjsonable(x) for x in v]
    return v


def mam_key(sol):
    return [[m.x, m.y, m.c, m.d, m.val, m.j, m.min_segment_length, str(m.mam_id), m.ref_chr_id] for m in sol]


def simulate_transcript_reads(exons, refs, rng, n):
    by_tr = {}
    for e in exons:
        by_tr.setdefault((e.seqid, e.transcript_id), []).append((e.start - 1, e.stop))
    keys = sorted(by_tr)
    reads = []
    for i in range(n):
        seqid, tid = keys[int(rng.randint(len(keys)))]
        spans = sorted(by_tr[(seqid, tid)])
        rna = "".join(refs[seqid][a:b] for a, b in spans).upper()
        ln = int(rng.randint(250, 1100))
        a = int(rng.randint(0, max(1, len(rna) - ln)))
        seq = mutate(rng, rna[a:a + ln], float(rng.choice([0.0, 0.03, 0.08, 0.14])))
        if i % 9 == 4:
            seq = rand_seq(rng, int(rng.randint(200, 500)))             # junk: unaligned
        if i % 4 == 1:
            seq = seq + "A" * int(rng.randint(12, 40))                           # poly-A tail
        if i % 2:
            seq = seq.translate(str.maketrans('ACGT', 'TGCA'))[::-1]
        qual = "".join(chr(33 + int(q)) for q in rng.randint(2, 40, len(seq))) if i % 3 == 0 else None
        reads.append(("sim%d_%s" % (i, tid), seq, qual))
    return reads


# ---- picklable stand-ins for the stages behind the MAM stage (the pc.main test starts consumer PROCESSES) ----
def stub_find_exons(chr_id, mam_solution, ref_exon_sequences, ref_segment_sequences, ref_flank_sequences, splice_pairs):
    from array import array
    exons = [(m.x, m.y, m.c, m.d, m.ref_chr_id) for m in mam_solution]
    seq = "".join(ref_segment_sequences.get(array('L', [m.ref_chr_id, m.x, m.y]).tobytes(), "") for m in mam_solution)
    pred = [(m.x, m.y) for m in mam_solution]
    splices = [(a[1], b[0]) for a, b in zip(pred[:-1], pred[1:])]
    return exons, seq, pred, splices, sum(m.d - m.c + 1 for m in mam_solution)


def stub_classify(chr_id, splices, *rest):
    return ("NO_SPLICE", "") if not splices else ("FSM" if len(splices) % 2 else "NIC_novel", "tr%d" % len(splices))


def stub_sam_entry(*a, **kw):
    return "\t".join(str(v) for v in a) + "\t" + str(sorted(kw.items())) + "\n"
