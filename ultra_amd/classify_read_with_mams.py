"""Drop-in for the reference's MAM stage, backed by libultra_hip.so.

Mirrors /root/reference/modules/classify_read_with_mams.py:
    main(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
         gene_to_small_segments, read_seq, warning_log_file, min_acc)
        -> (non_covered_regions, value, mam_solution)                        (:406-594, caller align.py:487)
    add_segment_to_mam(read_seq, ref_chr_id, exon_seq, e_start, e_stop, segm_id, mam_instance, min_acc, annot_label)
                                                                               (:303-402)
plus `main_batch`, which is what the restaged align_single calls: every (read, location) instance
of a batch is planned on the host (dict/set logic whose iteration order fixes tie order, kept in
Python exactly as SURVEY.md section 7.3 item 3 asks) and packed into flat arrays; ONE call of the
fused C-ABI entry point (ultra_mam_stage/_run/_fetch) then does everything else on the GPU: all
Myers searches (phase-B partial candidates speculatively), MAM materialisation, filters, the sort
by y and the chaining.  Only the chosen chains come back.

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
    """Flat (query, target, k) pair list over a byte arena; identical sequences and identical pairs
    are stored once."""

    def __init__(self):
        self.chunks, self.size, self.where, self.pair_of = [], 0, {}, {}
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
        qo, to, k = self.put(query), self.put(target), int(k)
        key = (qo, len(query), to, len(target), k)
        hit = self.pair_of.get(key)
        if hit is not None:
            return hit
        self.q_off.append(qo); self.q_len.append(len(query))
        self.t_off.append(to); self.t_len.append(len(target))
        self.k.append(k)
        self.pair_of[key] = len(self.k) - 1
        return len(self.k) - 1

    def arena(self):
        return np.frombuffer(b"".join(self.chunks), dtype=np.uint8)


def _queue_pairs(pairs, read, cand):
    """The edlib calls add_segment_to_mam would make for this candidate (:304-377)."""
    width = cand.e_stop - cand.e_start
    seq = cand.seq
    if len(seq) == 0 or len(read) == 0:
        return
    if cand.kind >= PART_SEG_START and len(seq) <= 5:
        return  # partial pieces of at most 5 bases are never aligned (:488, :497, :514, :520)
    if width >= 5:
        if width >= 9:
            cand.pair = pairs.add(seq, read, 0.4 * min(len(read), len(seq)))
        else:
            cand.pair = pairs.add(seq, read, 0)
    if width >= 0.8 * len(read):
        cand.pair_contained = pairs.add(read, seq, 0.4 * min(len(read), len(seq)))


class Batch(object):
    """A planned batch: the candidate plans (host objects, needed to name the MAMs afterwards) and the
    flat arrays of ultra_mam_input."""
    __slots__ = ("plans", "packed", "n_pairs", "n_inst", "cells", "algorithmic_bytes")


def pack_plans(plans):
    """Flattens candidate plans into the arrays of ultra_mam_input (include/ultra_hip.h)."""
    pairs = PairList()
    kind, e0, e1, pr, prc, sl, sid, coff, rl, fps, lps = [], [], [], [], [], [], [], [0], [], [], []
    for plan in plans:
        ids = {}
        for cand in plan.cands:
            cand.pair = cand.pair_contained = -1
            _queue_pairs(pairs, plan.read, cand)
            kind.append(cand.kind); e0.append(cand.e_start); e1.append(cand.e_stop)
            pr.append(cand.pair); prc.append(cand.pair_contained)
            sl.append(len(cand.seq)); sid.append(ids.setdefault(cand.seq, len(ids)))
        coff.append(len(kind))
        rl.append(len(plan.read)); fps.append(plan.first_part_stop); lps.append(plan.last_part_start)
    b = Batch()
    b.plans = plans
    b.packed = {
        "arena": pairs.arena() if pairs.size else np.zeros(1, np.uint8),
        "q_off": np.asarray(pairs.q_off, np.int64), "q_len": np.asarray(pairs.q_len, np.int32),
        "t_off": np.asarray(pairs.t_off, np.int64), "t_len": np.asarray(pairs.t_len, np.int32),
        "k": np.asarray(pairs.k, np.int32),
        "cand_kind": np.asarray(kind, np.uint8), "cand_e_start": np.asarray(e0, np.int32), "cand_e_stop": np.asarray(e1, np.int32),
        "cand_pair": np.asarray(pr, np.int32), "cand_pair_contained": np.asarray(prc, np.int32),
        "cand_seq_len": np.asarray(sl, np.int32), "cand_seq_id": np.asarray(sid, np.int32),
        "inst_cand_off": np.asarray(coff, np.int64), "inst_read_len": np.asarray(rl, np.int32),
        "inst_first_part_stop": np.asarray(fps, np.int64), "inst_last_part_start": np.asarray(lps, np.int64),
    }
    b.n_pairs, b.n_inst = len(pairs.k), len(plans)
    ql, tl = b.packed["q_len"].astype(np.int64), b.packed["t_len"].astype(np.int64)
    b.cells = int(np.dot(ql, tl)) if b.n_pairs else 0
    # m + n sequence bytes + 24 B descriptor + 16 B result header per pair
    b.algorithmic_bytes = int(ql.sum() + tl.sum() + 40 * b.n_pairs)
    return b


def prepare_batch(instances, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
                  gene_to_small_segments):
    """Host packing of a batch of (solution, read_seq) instances: candidate enumeration + flat arrays."""
    return pack_plans([plan_instance(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments,
                                     segment_to_gene, gene_to_small_segments, read_seq)
                       for solution, read_seq in instances])


def prepare_batch_native(index, instances):
    """The same batch built inside libultra_hip.so from a packed index (ultra_amd.index.NativeIndex, SURVEY.md
    section 8 row f4): no Python object per candidate."""
    return index.plan(instances)


def stage_batch(ctx, batch):
    """Host -> HBM copy of the packed batch (once; run_staged may repeat)."""
    if hasattr(batch, "handle"):
        ctx.mam_stage_plan(batch)
    else:
        ctx.mam_stage(batch.packed)


def run_staged(ctx, min_acc):
    """One device-only pass of the hot path over the staged batch (Myers -> MAMs -> chains)."""
    ctx.mam_run(min_acc)


def _mam_tuple(plan, res, prefix, r, j, cand_base):
    cand = plan.cands[int(res[prefix + "cand"][r]) - cand_base]
    msl = int(res[prefix + "msl"][r])
    val = float(res[prefix + "val"][r])
    if msl == 0:  # 5-8 bp exact segment: the reference stores the int len(exon_seq) (:357-367)
        val = int(val)
    return mam(int(res[prefix + "x"][r]), int(res[prefix + "y"][r]), int(res[prefix + "c"][r]), int(res[prefix + "d"][r]),
               val, j, msl, str(cand.ident) + LABELS[cand.kind], cand.chr_id)


def results_to_python(batch, res):
    """(non_covered_regions, value, mam_solution) per instance from the fetched arrays (:577-594).
    The j of a chain element is its index in the instance's sorted MAM list, as in the reference.
    Works on plain lists (one .tolist() per array): indexing numpy scalars one by one costs several times more."""
    out = []
    native = hasattr(batch, "handle")
    coff = batch.packed["inst_cand_off"].tolist()
    n_mam, chain_off, value = res["n_mam"].tolist(), res["chain_off"].tolist(), res["value"].tolist()
    cx, cy, cc, cd = res["chain_x"].tolist(), res["chain_y"].tolist(), res["chain_c"].tolist(), res["chain_d"].tolist()
    cval, cmsl, ccand, cj = res["chain_val"].tolist(), res["chain_msl"].tolist(), res["chain_cand"].tolist(), res["chain_j"].tolist()
    if native:
        kinds, chrs = batch.packed["cand_kind"].tolist(), batch.cand_chr.tolist()
    for i in range(batch.n_inst):
        if n_mam[i] == 0:
            out.append(([], -1, tuple()))
            continue
        picked = []
        for r in range(chain_off[i], chain_off[i + 1]):
            ci, msl = ccand[r], cmsl[r]
            val = int(cval[r]) if msl == 0 else cval[r]  # 5-8 bp exact segment: the reference stores the int len(exon_seq) (:357-367)
            if native:
                name, chr_id = batch.mam_id(ci, LABELS[kinds[ci]]), chrs[ci]
            else:
                cand = batch.plans[i].cands[ci - coff[i]]
                name, chr_id = str(cand.ident) + LABELS[cand.kind], cand.chr_id
            picked.append(mam(cx[r], cy[r], cc[r], cd[r], val, cj[r], msl, name, chr_id))
        read_len = batch.reads[i] if native else len(batch.plans[i].read)
        gaps = []
        if picked:
            gaps.append(picked[0].c)
            for p, q in zip(picked[:-1], picked[1:]):
                gaps.append(q.c - p.d - 1)
            gaps.append(read_len - picked[-1].d)
        out.append((gaps, value[i], tuple(picked)))
    return out


_SEG_NAMES = {}


def mam_name(kind, chr_id, e_start, e_stop):
    """mam_id of a chain element from what a want_names fetch returns: str(segm_id) + label for segments
    (segm_id = array('L', [chr, start, stop]).tobytes(), :445), "flank_<start>_<stop>" + label for flanks (:456)."""
    key = (kind, chr_id, e_start, e_stop)
    name = _SEG_NAMES.get(key)
    if name is None:
        if len(_SEG_NAMES) > 2000000:
            _SEG_NAMES.clear()
        if kind in (1, 4, 5):
            name = "flank_{0}_{1}".format(e_start, e_stop) + LABELS[kind]
        else:
            name = str(array('L', [chr_id, e_start, e_stop]).tobytes()) + LABELS[kind]
        _SEG_NAMES[key] = name
    return name


def results_from_names(res, read_lens):
    """(non_covered_regions, value, mam_solution) per instance of a device-planned batch, from a want_names fetch
    (the candidate arrays never left the GPU)."""
    out = []
    n_mam, chain_off, value = res["n_mam"].tolist(), res["chain_off"].tolist(), res["value"].tolist()
    cx, cy, cc, cd = res["chain_x"].tolist(), res["chain_y"].tolist(), res["chain_c"].tolist(), res["chain_d"].tolist()
    cval, cmsl, cj = res["chain_val"].tolist(), res["chain_msl"].tolist(), res["chain_j"].tolist()
    ck, cchr = res["chain_kind"].tolist(), res["chain_chr"].tolist()
    for i in range(len(n_mam)):
        if n_mam[i] == 0:
            out.append(([], -1, tuple()))
            continue
        picked = []
        for r in range(chain_off[i], chain_off[i + 1]):
            msl = cmsl[r]
            val = int(cval[r]) if msl == 0 else cval[r]  # 5-8 bp exact segment: the reference stores the int len(exon_seq) (:357-367)
            picked.append(mam(cx[r], cy[r], cc[r], cd[r], val, cj[r], msl, mam_name(ck[r], cchr[r], cx[r], cy[r]), cchr[r]))
        gaps = []
        if picked:
            gaps.append(picked[0].c)
            for p, q in zip(picked[:-1], picked[1:]):
                gaps.append(q.c - p.d - 1)
            gaps.append(read_lens[i] - picked[-1].d)
        out.append((gaps, value[i], tuple(picked)))
    return out


def mam_lists_to_python(batch, res):
    """Every instance's complete MAM list (sorted by y, j = position), from a want_all fetch."""
    coff = batch.packed["inst_cand_off"]
    out = []
    for i, plan in enumerate(batch.plans):
        a, b = int(res["mam_off"][i]), int(res["mam_off"][i + 1])
        out.append([_mam_tuple(plan, res, "mam_", r, r - a, int(coff[i])) for r in range(a, b)])
    return out


def main_batch(instances, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
               gene_to_small_segments, min_acc, ctx=None, stats=None, index=None):
    """instances: iterable of (solution, read_seq). Returns one (non_covered_regions, value,
    mam_solution) per instance, equal to what the reference's main() returns for each.
    index: an ultra_amd.index.NativeIndex built once from the same five dictionaries - the batch is then planned
    natively (the dictionaries themselves are not read)."""
    ctx = ctx or _lib.default_context()
    if index is not None:
        batch = prepare_batch_native(index, list(instances))
    else:
        batch = prepare_batch(instances, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
                              gene_to_small_segments)
    if batch.n_inst == 0:
        return []
    stage_batch(ctx, batch)
    run_staged(ctx, min_acc)
    res = ctx.mam_fetch()
    if stats is not None:
        stats["pairs"] = stats.get("pairs", 0) + batch.n_pairs
        stats["cells"] = stats.get("cells", 0) + batch.cells
        stats["mams"] = stats.get("mams", 0) + int(res["n_mam"].sum())
    return results_to_python(batch, res)


def main(solution, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene,
         gene_to_small_segments, read_seq, warning_log_file, min_acc):
    """Single-instance form with the reference's signature (:406). One GPU batch per call: for
    throughput use main_batch."""
    return main_batch([(solution, read_seq)], ref_segment_sequences, ref_flank_sequences, parts_to_segments,
                      segment_to_gene, gene_to_small_segments, min_acc)[0]


def add_segment_to_mam(read_seq, ref_chr_id, exon_seq, e_start, e_stop, segm_id, mam_instance, min_acc, annot_label,
                       ctx=None):
    """Reference signature (:303): appends this candidate's MAMs to mam_instance (j = 0 / len(exon_seq)
    as the reference leaves it before main() renumbers).  One single-candidate GPU batch per call."""
    ctx = ctx or _lib.default_context()
    # a lone candidate is staged as kind 0 so that no instance-level filter applies; the label is only a name here
    cand = Candidate(FULL_SEGMENT, ref_chr_id, e_start, e_stop, exon_seq, segm_id, -1)
    plan = Plan()
    plan.read, plan.cands, plan.first_part_stop, plan.last_part_start = read_seq, [cand], 2 ** 32, 0
    batch = pack_plans([plan])
    stage_batch(ctx, batch)
    run_staged(ctx, min_acc)
    res = ctx.mam_fetch(want_all=True)
    name = str(segm_id) + annot_label
    for r in range(int(res["mam_off"][1])):
        msl = int(res["mam_msl"][r])
        val = float(res["mam_val"][r])
        if msl == 0:
            val = int(val)
        mam_instance.append(mam(int(res["mam_x"][r]), int(res["mam_y"][r]), int(res["mam_c"][r]), int(res["mam_d"][r]),
                                val, val if msl == 0 else 0, msl, name, ref_chr_id))
