"""CPU restatement of the reference's MAM stage (TEST INFRASTRUCTURE ONLY).

Nothing under ultra_amd/ may import this module; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do.

Restates, sequentially and per (read, location) instance:
  edlib_alignment / cigar_to_accuracy   /root/reference/modules/classify_read_with_mams.py:71-119
  add_segment_to_mam                     classify_read_with_mams.py:303-402
  get_unique_exon_and_flank_locations    classify_read_with_mams.py:180-253
  get_unique_segment_and_flank_choordinates  classify_read_with_mams.py:256-300
  main                                   classify_read_with_mams.py:406-594
The edlib call inside is oracle.cport.hw_align (see oracle/edlib_port.c for its parity status);
the chainers are oracle/chain_port.c.  Pinned in tests against golden vectors produced in the
build container by running the reference's own module with edlib bound to the same oracle
(tests/golden/make_classify_golden.py).
"""
from array import array
from collections import namedtuple

from . import cport

Mam = namedtuple('Mam', ['x', 'y', 'c', 'd', 'val', 'j', "min_segment_length", "mam_id", "ref_chr_id"])

SMALL_MIN = 5      # segments shorter than this are never aligned (:304)
EXACT_BELOW = 9    # 5..8 bp segments need an exact hit (:311, :352)
CONTAINED_FRAC = 0.8  # read-in-segment test (:374)
SWITCH_N = 200     # > SWITCH_N MAMs -> n log n solver (:561)


def _key(chr_id, a, b):
    return array('L', [chr_id, a, b]).tobytes()


def approx_match(query, target, k):
    """edlib_alignment(query, target, mode='HW', task='path', k) (:98-119)."""
    res = cport.hw_align(query, target, int(k))
    if res["editDistance"] == -1:
        return [0, 0], -1, 0
    first_s, first_e = res["locations"][0]
    span = first_e - first_s + 1
    ins = res["insertions"]
    # cigar_to_accuracy (:71-94): '=' columns over all columns of the first location's path
    matches = span - res["editDistance"] + ins
    return res["locations"], res["editDistance"], matches / float(span + ins)


def emit_mams(read, chr_id, seq, e_start, e_stop, ident, out, min_acc, label):
    """add_segment_to_mam (:303-402); appends to `out`."""
    width = e_stop - e_start
    name = str(ident) + label
    if width >= SMALL_MIN:
        if width >= EXACT_BELOW:
            locs, ed, acc = approx_match(seq, read, 0.4 * min(len(read), len(seq)))
            if ed >= 0 and acc > min_acc:
                seen, last_score = set(), 0
                for s, e in locs:
                    span = e - s + 1
                    score = max((span - ed) / span, acc) * span
                    if s in seen and score <= last_score:
                        continue
                    out.append(Mam(e_start, e_stop, s, e, score, 0, span, name, chr_id))
                    seen.add(s)
                    last_score = score
        else:
            locs, ed, acc = approx_match(seq, read, 0)
            if ed == 0:
                seen = set()
                for s, e in locs:
                    if s in seen:
                        continue
                    out.append(Mam(e_start, e_stop, s, e, len(seq), len(seq), 0, name, chr_id))
                    seen.add(s)
    if width >= CONTAINED_FRAC * len(read):
        locs, ed, acc = approx_match(read, seq, 0.4 * min(len(read), len(seq)))
        if ed >= 0:
            s, e = locs[0]
            span = e - s + 1
            if acc > min_acc:
                out.append(Mam(e_start, e_stop, 0, len(read) - 1, acc * span, 0, span, name, chr_id))


def _triples(flat):
    return [(flat[i], flat[i + 1], flat[i + 2]) for i in range(0, len(flat), 3)]


def _touches(a0, a1, b0, b1):
    return (a0 <= b0 <= a1) or (a0 <= b1 <= a1) or (b0 <= a0 <= a1 <= b1)


def plan_candidates(solution, parts_to_segments, segment_to_gene, gene_to_small_segments):
    """Candidate enumeration (:180-300), returning ordered dictionaries like the reference's."""
    hit_segments, hit_flanks = [], []
    seg_extent, flank_extent, seg_ident = {}, {}, {}
    first_part_stop, last_part_start = 2 ** 32, 0
    for nam in solution:
        chr_id, p0, p1 = (int(v) for v in nam.exon_part_id.split('^'))
        pkey = _key(chr_id, p0, p1)
        if pkey not in parts_to_segments:
            loc = (chr_id, p0, p1)
            hit_flanks.append(loc)
            if loc in flank_extent:
                flank_extent[loc][1] = nam.y
                flank_extent[loc][3] = nam.d
            else:
                flank_extent[loc] = [nam.x, nam.y, nam.c, nam.d]
            continue
        if p1 <= first_part_stop:
            first_part_stop = p1
        if p0 >= last_part_start:
            last_part_start = p0
        for _, s0, s1 in _triples(parts_to_segments[pkey]):
            if _touches(s0, s1, nam.x, nam.y):
                loc = (chr_id, s0, s1)
                seg_ident[loc] = _key(_, s0, s1)
                hit_segments.append(loc)
                if loc in seg_extent:
                    seg_extent[loc][1] = nam.y
                    seg_extent[loc][3] = nam.d
                else:
                    seg_extent[loc] = [nam.x, nam.y, nam.c, nam.d]
    hit_segments = sorted(set(hit_segments), key=lambda v: v[1])

    full_segments, partial_segments, full_flanks, partial_flanks = {}, {}, {}, {}
    if hit_segments:
        genes = set(g for loc in hit_segments for g in segment_to_gene[seg_ident[loc]])
        small = set(t for g in genes for t in _triples(gene_to_small_segments[g]))
        for c_id, s0, s1 in small:
            full_segments.setdefault((c_id, s0, s1), set()).add(_key(c_id, s0, s1))
    for loc in hit_segments:
        full_segments.setdefault(loc, set()).add(seg_ident[loc])
        if loc in seg_extent:
            g0, g1, _, _ = seg_extent[loc]
            partial_segments[loc] = (loc[0], g0, g1, seg_ident[loc])
    for loc in hit_flanks:
        full_flanks[loc] = set()
        g0, g1, _, _ = flank_extent[loc]
        width = loc[2] - loc[1]
        if (g0 - loc[1]) > 0.05 * width:
            partial_flanks[loc] = (loc[0], g0, g1)
        if (loc[2] - g1) > 0.05 * width:
            partial_flanks[loc] = (loc[0], g0, g1)
    return full_segments, partial_segments, full_flanks, partial_flanks, first_part_stop, last_part_start


def mam_instance(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
                 gene_to_small_segments, read_seq, min_acc):
    """The filtered MAM list of main, sorted by y with j renumbered (:419-555)."""
    full_segments, partial_segments, full_flanks, partial_flanks, first_part_stop, last_part_start = \
        plan_candidates(solution, parts_to_segments, segment_to_gene, gene_to_small_segments)

    mams = []
    for (chr_id, s0, s1), idents in sorted(full_segments.items(), key=lambda kv: kv[0][1]):
        emit_mams(read_seq, chr_id, ref_segment_sequences[_key(chr_id, s0, s1)], s0, s1, idents.pop(), mams,
                  min_acc, '_full_segment')
    for (chr_id, f0, f1), _ in sorted(full_flanks.items(), key=lambda kv: kv[0][1]):
        emit_mams(read_seq, chr_id, ref_flank_sequences[_key(chr_id, f0, f1)], f0, f1,
                  "flank_{0}_{1}".format(f0, f1), mams, min_acc, '_full_flank')

    if mams:
        lowest_stop = min(m.y for m in mams)
        highest_start = max(m.x for m in mams)
    else:
        lowest_stop, highest_start = -1, 2 ** 32
    start_exon_limit = max(first_part_stop, lowest_stop)
    end_exon_limit = min(last_part_start, highest_start)

    tried = set()
    for (chr_id, e0, e1), (_, g0, g1, ident) in partial_segments.items():
        seq = ref_segment_sequences[_key(chr_id, e0, e1)]
        if e1 <= start_exon_limit:
            piece, label = seq[g0 - e0:], '_partial_segment_start'
        elif end_exon_limit <= e0:
            piece, label = seq[: len(seq) - (e1 - (g1 + 1))], '_partial_segment_end'
        else:
            continue
        if piece not in tried and len(piece) > 5:
            emit_mams(read_seq, chr_id, piece, e0, e1, ident, mams, min_acc, label)
            tried.add(piece)

    tried = set()
    for (chr_id, f0, f1), (_, g0, g1) in partial_flanks.items():
        seq = ref_flank_sequences[_key(chr_id, f0, f1)]
        ident = "flank_{0}_{1}".format(f0, f1)
        for piece, label in ((seq[g0 - f0:], '_partial_flank_start'),
                             (seq[: len(seq) - (f1 - (g1 + 1))], '_partial_flank_end')):
            if piece not in tried and len(piece) > 5:
                emit_mams(read_seq, chr_id, piece, f0, f1, ident, mams, min_acc, label)
                tried.add(piece)

    # keep only the outermost partial hits (:530-547)
    p_start, p_end, min_d, max_c = 2 ** 32, 0, 2 ** 32, 0
    for m in mams:
        if "_start" in m.mam_id and m.c < p_start:
            p_start = m.c
        elif "_end" in m.mam_id and m.d > p_end:
            p_end = m.d
        min_d = min(min_d, m.d)
        max_c = max(max_c, m.c)
    p_start, p_end = min(p_start, min_d), max(p_end, max_c)
    mams = [m for m in mams
            if not ("_start" in m.mam_id and m.c > p_start) and not ("_end" in m.mam_id and m.d < p_end)]

    mams.sort(key=lambda m: m.y)
    return [m._replace(j=j) for j, m in enumerate(mams)]


def classify(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
             gene_to_small_segments, read_seq, warning_log_file, min_acc):
    """main (:406-594): returns (non_covered_regions, value, mam_solution)."""
    mams = mam_instance(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
                        gene_to_small_segments, read_seq, min_acc)
    if not mams:
        return [], -1, tuple()

    xs = [m.x for m in mams]; ys = [m.y for m in mams]
    cs = [m.c for m in mams]; ds = [m.d for m in mams]; vals = [m.val for m in mams]
    if len(mams) > SWITCH_N:
        chain, value = cport.chain_nlogn(xs, ys, cs, ds, vals, 5)
    else:
        chain, value = cport.chain_quadratic(cs, ds, vals, 20)
    picked = tuple(mams[j] for j in chain)
    gaps = []
    if picked:
        gaps.append(picked[0].c)
        for a, b in zip(picked[:-1], picked[1:]):
            gaps.append(b.c - a.d - 1)
        gaps.append(len(read_seq) - picked[-1].d)
    return gaps, value, picked
