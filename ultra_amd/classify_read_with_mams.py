"""Drop-in for the reference's MAM stage, backed by libultra_hip.so.

Mirrors /root/reference/modules/classify_read_with_mams.py:
    main(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
         gene_to_small_segments, read_seq, warning_log_file, min_acc)
        -> (non_covered_regions, value, mam_solution)                        (:406-594, caller align.py:487)
    add_segment_to_mam(read_seq, ref_chr_id, exon_seq, e_start, e_stop, segm_id, mam_instance, min_acc, annot_label)
                                                                               (:303-402)
plus `main_batch`, which is what the restaged align_single calls: every (read, location) instance
of a batch is planned on the host (dict/set logic whose iteration order fixes tie order, kept in
Python exactly as SURVEY.md section 7.3 item 3 asks), all candidate (segment, read) pairs of all
instances go to the GPU in ONE Myers batch (phase-B partial candidates speculatively), the MAM
lists are materialised from the returned distances/locations, and all instances are chained in ONE
GPU chaining batch.

There is no CPU fallback: without libultra_hip.so / a gfx950 device every call raises.
"""
from array import array
from collections import namedtuple

import numpy as np

from . import lib as _lib

mam = namedtuple('Mam', ['x', 'y', 'c', 'd', 'val', 'j', "min_segment_length", "mam_id", "ref_chr_id"])
globals()[mam.__name__] = mam

# candidate kinds, in the order the reference emits them
FULL_SEGMENT, FULL_FLANK, PART_SEG_START, PART_SEG_END, PART_FLANK_START, PART_FLANK_END = range(6)
LABELS = ('_full_segment', '_full_flank', '_partial_segment_start', '_partial_segment_end',
          '_partial_flank_start', '_partial_flank_end')
_IS_START = (False, False, True, False, True, False)
_IS_END = (False, False, False, True, False, True)


def _key(chr_id, a, b):
    return array('L', [chr_id, a, b]).tobytes()


def _grouped3(flat):
    it = iter(flat)
    return zip(it, it, it)


def _overlaps(a0, a1, b0, b1):
    a0, a1, b0, b1 = int(a0), int(a1), int(b0), int(b1)
    return (a0 <= b0 <= a1) or (a0 <= b1 <= a1) or (b0 <= a0 <= a1 <= b1)


class Candidate(object):
    __slots__ = ("kind", "chr_id", "e_start", "e_stop", "seq", "ident", "group", "pair", "pair_contained")

    def __init__(self, kind, chr_id, e_start, e_stop, seq, ident, group):
        self.kind, self.chr_id, self.e_start, self.e_stop = kind, chr_id, e_start, e_stop
        self.seq, self.ident, self.group = seq, ident, group
        self.pair = self.pair_contained = -1


class Plan(object):
    """Ordered candidate list of one (read, location) instance + the two part limits (:194-221)."""
    __slots__ = ("read", "cands", "first_part_stop", "last_part_start")


def plan_instance(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
                  gene_to_small_segments, read_seq):
    """Host-side candidate enumeration (:180-300 and the loops of :441-523), flattened into one ordered
    list. Partial candidates are listed speculatively; `group` ties the two variants of one partial
    segment together (the reference takes the end variant only in the `elif`, :485-500)."""
    seg_hits, flank_hits = [], []
    seg_span, flank_span, seg_id = {}, {}, {}
    first_part_stop, last_part_start = 2 ** 32, 0
    for nam in solution:
        f = nam.exon_part_id.split('^')
        chr_id, p0, p1 = int(f[0]), int(f[1]), int(f[2])
        pk = _key(chr_id, p0, p1)
        if pk in parts_to_segments:
            if p1 <= first_part_stop:
                first_part_stop = p1
            if p0 >= last_part_start:
                last_part_start = p0
            for t_chr, s0, s1 in _grouped3(parts_to_segments[pk]):
                if _overlaps(s0, s1, nam.x, nam.y):
                    loc = (chr_id, s0, s1)
                    seg_id[loc] = _key(t_chr, s0, s1)
                    seg_hits.append(loc)
                    span = seg_span.get(loc)
                    if span is None:
                        seg_span[loc] = [nam.x, nam.y, nam.c, nam.d]
                    else:
                        span[1], span[3] = nam.y, nam.d
        else:
            loc = (chr_id, p0, p1)
            flank_hits.append(loc)
            span = flank_span.get(loc)
            if span is None:
                flank_span[loc] = [nam.x, nam.y, nam.c, nam.d]
            else:
                span[1], span[3] = nam.y, nam.d
    seg_hits = list(set(seg_hits))
    seg_hits.sort(key=lambda v: v[1])

    full = {}  # coordinate -> identifier, insertion-ordered like the reference's defaultdict
    if seg_hits:
        genes = set(g for loc in seg_hits for g in segment_to_gene[seg_id[loc]])
        small = set(t for g in genes for t in _grouped3(gene_to_small_segments[g]))
        for t_chr, s0, s1 in small:
            full[(t_chr, s0, s1)] = _key(t_chr, s0, s1)
    partial_segs = {}
    for loc in seg_hits:
        if loc not in full:
            full[loc] = seg_id[loc]
        if loc in seg_span:
            partial_segs[loc] = (seg_span[loc][0], seg_span[loc][1], seg_id[loc])
    flanks, partial_flanks = {}, {}
    for loc in flank_hits:
        flanks[loc] = True
        g0, g1 = flank_span[loc][0], flank_span[loc][1]
        width = loc[2] - loc[1]
        if (g0 - loc[1]) > 0.05 * width or (loc[2] - g1) > 0.05 * width:
            partial_flanks[loc] = (g0, g1)

    cands = []
    for loc in sorted(full, key=lambda v: v[1]):
        cands.append(Candidate(FULL_SEGMENT, loc[0], loc[1], loc[2], ref_segment_sequences[_key(*loc)], full[loc], -1))
    for loc in sorted(flanks, key=lambda v: v[1]):
        cands.append(Candidate(FULL_FLANK, loc[0], loc[1], loc[2], ref_flank_sequences[_key(*loc)],
                               "flank_{0}_{1}".format(loc[1], loc[2]), -1))
    group = 0
    for loc, (g0, g1, ident) in partial_segs.items():
        seq = ref_segment_sequences[_key(*loc)]
        cands.append(Candidate(PART_SEG_START, loc[0], loc[1], loc[2], seq[g0 - loc[1]:], ident, group))
        cands.append(Candidate(PART_SEG_END, loc[0], loc[1], loc[2], seq[: len(seq) - (loc[2] - (g1 + 1))], ident, group))
        group += 1
    for loc, (g0, g1) in partial_flanks.items():
        seq = ref_flank_sequences[_key(*loc)]
        ident = "flank_{0}_{1}".format(loc[1], loc[2])
        cands.append(Candidate(PART_FLANK_START, loc[0], loc[1], loc[2], seq[g0 - loc[1]:], ident, group))
        cands.append(Candidate(PART_FLANK_END, loc[0], loc[1], loc[2], seq[: len(seq) - (loc[2] - (g1 + 1))], ident, group))
        group += 1
    p = Plan()
    p.read, p.cands, p.first_part_stop, p.last_part_start = read_seq, cands, first_part_stop, last_part_start
    return p


class PairList(object):
    """Flat (query, target, k) pair list over a byte arena, deduplicating identical sequences."""

    def __init__(self):
        self.chunks, self.size, self.where = [], 0, {}
        self.q_off, self.q_len, self.t_off, self.t_len, self.k = [], [], [], [], []

    def put(self, seq):
        hit = self.where.get(seq)
        if hit is not None:
            return hit
        b = seq.encode() if isinstance(seq, str) else bytes(seq)
        off = self.size
        self.chunks.append(b)
        self.size += len(b)
        self.where[seq] = off
        return off

    def add(self, query, target, k):
        self.q_off.append(self.put(query)); self.q_len.append(len(query))
        self.t_off.append(self.put(target)); self.t_len.append(len(target))
        self.k.append(int(k))
        return len(self.k) - 1

    def arena(self):
        return np.frombuffer(b"".join(self.chunks), dtype=np.uint8)


def _queue_pairs(pairs, read, cand):
    """The edlib calls add_segment_to_mam would make for this candidate (:304-377)."""
    width = cand.e_stop - cand.e_start
    seq = cand.seq
    if len(seq) == 0 or len(read) == 0:
        return
    if width >= 5:
        if width >= 9:
            cand.pair = pairs.add(seq, read, 0.4 * min(len(read), len(seq)))
        else:
            cand.pair = pairs.add(seq, read, 0)
    if width >= 0.8 * len(read):
        cand.pair_contained = pairs.add(read, seq, 0.4 * min(len(read), len(seq)))


def _accuracy(res, p):
    """edlib_alignment + cigar_to_accuracy (:71-119) from (first location, ed, insertions)."""
    lo = int(res["loc_off"][p])
    span = int(res["loc_end"][lo]) - int(res["loc_start"][lo]) + 1
    ins = int(res["first_ins"][p])
    return (span - int(res["ed"][p]) + ins) / float(span + ins)


def _emit(res, read, cand, out, min_acc):
    """add_segment_to_mam (:303-402) on GPU results."""
    width = cand.e_stop - cand.e_start
    name = str(cand.ident) + LABELS[cand.kind]
    p = cand.pair
    if p >= 0:
        ed = int(res["ed"][p])
        lo, hi = int(res["loc_off"][p]), int(res["loc_off"][p + 1])
        if width >= 9:
            if ed >= 0:
                acc = _accuracy(res, p)
                if acc > min_acc:
                    seen, last_score = set(), 0
                    for a in range(lo, hi):
                        s, e = int(res["loc_start"][a]), int(res["loc_end"][a])
                        span = e - s + 1
                        score = max((span - ed) / span, acc) * span
                        if s in seen and score <= last_score:
                            continue
                        out.append(mam(cand.e_start, cand.e_stop, s, e, score, 0, span, name, cand.chr_id))
                        seen.add(s)
                        last_score = score
        elif ed == 0:
            seen = set()
            n = len(cand.seq)
            for a in range(lo, hi):
                s, e = int(res["loc_start"][a]), int(res["loc_end"][a])
                if s in seen:
                    continue
                out.append(mam(cand.e_start, cand.e_stop, s, e, n, n, 0, name, cand.chr_id))
                seen.add(s)
    p = cand.pair_contained
    if p >= 0 and int(res["ed"][p]) >= 0:
        acc = _accuracy(res, p)
        lo = int(res["loc_off"][p])
        span = int(res["loc_end"][lo]) - int(res["loc_start"][lo]) + 1
        if acc > min_acc:
            out.append(mam(cand.e_start, cand.e_stop, 0, len(read) - 1, acc * span, 0, span, name, cand.chr_id))


def _materialise(plan, res, min_acc):
    """Phase A, the first/last-exon reduction, phase B with its de-duplication, the outermost-partial
    filter and the stable sort by y (:440-555)."""
    read = plan.read
    mams = []
    cands = plan.cands
    i = 0
    while i < len(cands) and cands[i].kind in (FULL_SEGMENT, FULL_FLANK):
        _emit(res, read, cands[i], mams, min_acc)
        i += 1
    if mams:
        first_stop = max(plan.first_part_stop, min(m.y for m in mams))
        last_start = min(plan.last_part_start, max(m.x for m in mams))
    else:
        first_stop = max(plan.first_part_stop, -1)
        last_start = min(plan.last_part_start, 2 ** 32)
    tried = set()
    while i < len(cands) and cands[i].kind in (PART_SEG_START, PART_SEG_END):
        a, b = cands[i], cands[i + 1]
        i += 2
        pick = a if a.e_stop <= first_stop else (b if last_start <= b.e_start else None)
        if pick is not None and pick.seq not in tried and len(pick.seq) > 5:
            _emit(res, read, pick, mams, min_acc)
            tried.add(pick.seq)
    tried = set()
    while i < len(cands):
        pick = cands[i]
        i += 1
        if pick.seq not in tried and len(pick.seq) > 5:
            _emit(res, read, pick, mams, min_acc)
            tried.add(pick.seq)

    p_start, p_end, min_d, max_c = 2 ** 32, 0, 2 ** 32, 0
    for m in mams:
        if "_start" in m.mam_id and m.c < p_start:
            p_start = m.c
        elif "_end" in m.mam_id and m.d > p_end:
            p_end = m.d
        if m.d < min_d:
            min_d = m.d
        if m.c > max_c:
            max_c = m.c
    p_start, p_end = min(p_start, min_d), max(p_end, max_c)
    kept = [m for m in mams if not ("_start" in m.mam_id and m.c > p_start) and not ("_end" in m.mam_id and m.d < p_end)]
    kept.sort(key=lambda m: m.y)
    return [m._replace(j=j) for j, m in enumerate(kept)]


def main_batch(instances, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
               gene_to_small_segments, min_acc, ctx=None, stats=None):
    """instances: iterable of (solution, read_seq). Returns one (non_covered_regions, value,
    mam_solution) per instance, equal to what the reference's main() returns for each."""
    ctx = ctx or _lib.default_context()
    plans, pairs = [], PairList()
    for solution, read_seq in instances:
        plan = plan_instance(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments,
                             segment_to_gene, gene_to_small_segments, read_seq)
        for cand in plan.cands:
            _queue_pairs(pairs, read_seq, cand)
        plans.append(plan)
    if pairs.k:
        res = ctx.myers_hw_batch(pairs.arena(), pairs.q_off, pairs.q_len, pairs.t_off, pairs.t_len, pairs.k)
    else:
        res = None
    lists = [_materialise(plan, res, min_acc) for plan in plans]
    off = np.zeros(len(lists) + 1, np.int64)
    np.cumsum([len(l) for l in lists], out=off[1:])
    flat = [m for l in lists for m in l]
    if flat:
        idx, ln, value = ctx.chain_mams_batch(off, [m.x for m in flat], [m.y for m in flat], [m.c for m in flat],
                                              [m.d for m in flat], [float(m.val) for m in flat])
    if stats is not None:
        stats["pairs"] = stats.get("pairs", 0) + len(pairs.k)
        stats["cells"] = stats.get("cells", 0) + (int(np.dot(np.asarray(pairs.q_len, np.int64), np.asarray(pairs.t_len, np.int64))) if pairs.k else 0)
        stats["mams"] = stats.get("mams", 0) + len(flat)
    out = []
    for i, (plan, l) in enumerate(zip(plans, lists)):
        if not l:
            out.append(([], -1, tuple()))
            continue
        o = int(off[i])
        picked = tuple(l[j] for j in idx[o:o + int(ln[i])])
        gaps = []
        if picked:
            gaps.append(picked[0].c)
            for a, b in zip(picked[:-1], picked[1:]):
                gaps.append(b.c - a.d - 1)
            gaps.append(len(plan.read) - picked[-1].d)
        out.append((gaps, float(value[i]), picked))
    return out


def main(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
         gene_to_small_segments, read_seq, warning_log_file, min_acc):
    """Single-instance form with the reference's signature (:406). One GPU batch per call: for
    throughput use main_batch."""
    return main_batch([(solution, read_seq)], ref_segment_sequences, ref_flank_sequences, parts_to_segments,
                      segment_to_gene, gene_to_small_segments, min_acc)[0]


def add_segment_to_mam(read_seq, ref_chr_id, exon_seq, e_start, e_stop, segm_id, mam_instance, min_acc, annot_label):
    """Reference signature (:303): appends this candidate's MAMs to mam_instance."""
    cand = Candidate(LABELS.index(annot_label), ref_chr_id, e_start, e_stop, exon_seq, segm_id, -1)
    pairs = PairList()
    _queue_pairs(pairs, read_seq, cand)
    if not pairs.k:
        return
    res = _lib.default_context().myers_hw_batch(pairs.arena(), pairs.q_off, pairs.q_len, pairs.t_off, pairs.t_len, pairs.k)
    _emit(res, read_seq, cand, mam_instance, min_acc)
