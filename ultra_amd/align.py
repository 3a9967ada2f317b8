"""Host side of the steps right before the MAM stage, batched: what
/root/reference/modules/align.py:427-489 does per read - parse the seeds, split them into per-chromosome
instances with an upper bound, chain the NAMs of every instance, pick the locations worth a MAM stage - with the
two chaining calls of :448-453 replaced by ONE ultra_chain_nams_batch launch for the whole batch and the MAM
stage by ONE ultra_mam_* pass (classify_read_with_mams.main_batch).

The reference stops chaining the instances of a read once an instance's upper bound drops below
dropoff * best value so far (:443).  Here every instance is chained on the GPU first and the same rule is applied
afterwards in the same order, which selects exactly the same set.
"""
from collections import defaultdict, namedtuple

from . import classify_read_with_mams, colinear_solver

mem = namedtuple('Mem', ['x', 'y', 'c', 'd', 'val', 'j', "exon_part_id"])

# IUPAC complement, both cases; U pairs with A; X stays X (upper case only, as in the reference's table)
_IUPAC_FROM = "ACGTUNRYKMSWBVDH"
_IUPAC_TO = "TGCAANYRMKSWVBHD"
_COMP = str.maketrans(_IUPAC_FROM + _IUPAC_FROM.lower() + "X", _IUPAC_TO + _IUPAC_TO.lower() + "X")
_KNOWN = frozenset(_IUPAC_FROM + _IUPAC_FROM.lower() + "X")


def reverse_complement(seq):
    """help_functions.py:93-97: IUPAC-aware reverse complement; a character outside the table is a KeyError there
    and here."""
    bad = set(seq) - _KNOWN
    if bad:
        raise KeyError(sorted(bad)[0])
    return seq.translate(_COMP)[::-1]


def get_mems_from_input(hits):
    """Hit lines -> {chr_id: [Mem sorted by y, j = rank]}  (align.py:357-388)."""
    rows = defaultdict(list)
    for line in hits:
        vals = line.split()
        part_id = vals[0]
        chr_id, part_start, _ = part_id.split('^')
        n = int(vals[3])
        x = int(part_start) + int(vals[1]) - 1
        c = int(vals[2]) - 1
        rows[int(chr_id)].append((x, x + n - 1, c, c + n - 1, n, part_id))
    out = defaultdict(list)
    for chr_id, tuples in rows.items():
        tuples.sort(key=lambda t: t[1])       # stable, like the reference's sorted()
        out[chr_id] = [mem(x, y, c, d, val, j, pid) for j, (x, y, c, d, val, pid) in enumerate(tuples)]
    return out


def _covered_bases(mems):
    """Size of the union of the inclusive read intervals [c, d] (the sweep of align.py:157-182)."""
    total, end = 0, -1
    for c, d in sorted((m.c, m.d) for m in mems):
        if d > end:
            total += d - max(c, end + 1) + 1
            end = d
    return total


def annotate_guaranteed_optimal_bound(mems, is_rc, max_intron_chr, max_global_intron):
    """{(chr_id, instance index): (upper bound, is_rc, mems of the instance re-indexed from 0)}: the seeds of a
    chromosome are cut wherever two consecutive ones are more than the allowed intron apart (align.py:112-188)."""
    upper_bound = {}
    for chr_id, chr_mems in mems.items():
        limit = min(max_intron_chr[chr_id] + 20000, max_global_intron)
        if len(chr_mems) > 1:
            inst, idx = [chr_mems[0]._replace(j=0)], 0
            for m1, m2 in zip(chr_mems[:-1], chr_mems[1:]):
                if m2.x - m1.y > limit:
                    upper_bound[(chr_id, idx)] = (_covered_bases(inst), is_rc, inst)
                    idx += 1
                    inst = [m2._replace(j=0)]
                else:
                    inst.append(m2._replace(j=len(inst)))
            upper_bound[(chr_id, idx)] = (_covered_bases(inst), is_rc, inst)
        else:
            upper_bound[(chr_id, 0)] = (_covered_bases(chr_mems), is_rc, chr_mems)
    return upper_bound


def chain_reads_batch(seed_records, max_intron_chr, max_global_intron, dropoff, max_loc, ctx=None, last_limits=None):
    """seed_records: [(hits, hits_rc)] per read (raw hit lines, seed_wrapper.read_seeds).  Returns per read the
    locations the reference would hand to the MAM stage, in its order: [(chr_id, mem_solution, score, is_rc)]
    (align.py:429-475, before the duplicate-MAM-solution filter of :493 which needs the MAM results).
    last_limits, if a list, receives per read the intron limit of the LAST instance the reference's chaining loop
    worked on (its variable max_allowed_intron as the loop leaves it, :446) - the value its later intron filter
    (:505) compares every location of the read with, whatever chromosome the location is on."""
    per_read, flat, introns = [], [], []
    for hits, hits_rc in seed_records:
        ub = annotate_guaranteed_optimal_bound(get_mems_from_input(hits), False, max_intron_chr, max_global_intron)
        ub_rc = annotate_guaranteed_optimal_bound(get_mems_from_input(hits_rc), True, max_intron_chr, max_global_intron)
        order = sorted(list(ub.items()) + list(ub_rc.items()), key=lambda kv: kv[1][0], reverse=True)
        per_read.append((order, len(flat)))
        for (chr_id, _), (_, _, inst) in order:
            flat.append(inst)
            introns.append(min(max_intron_chr[chr_id] + 20000, max_global_intron))
    chained = colinear_solver.read_coverage_batch(flat, introns, solver=0, ctx=ctx) if flat else []
    out = []
    for order, base in per_read:
        all_chainings, best, limit = [], 0, None
        for k, ((chr_id, _), (bound, is_rc, _)) in enumerate(order):
            if bound < best * dropoff:
                break
            limit = introns[base + k]
            solutions, value = chained[base + k]
            best = max(best, value)
            for sol in solutions:
                all_chainings.append((chr_id, sol, value, is_rc))
        all_chainings.sort(key=lambda t: (-t[2], t[1][-1].y - t[1][0].x))
        kept = []
        for i, loc in enumerate(all_chainings):
            if loc[2] / float(all_chainings[0][2]) < dropoff or i >= max_loc:
                break
            kept.append(loc)
        out.append(kept)
        if last_limits is not None:
            last_limits.append(limit)
    return out


def mam_stage_batch(reads, locations, index, min_acc, ctx=None):
    """reads: [read_seq] (already poly-A trimmed, align.py:431); locations: chain_reads_batch's result; index =
    (ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene, gene_to_small_segments).
    One GPU MAM stage over every (read, location); returns per read [(chr_id, is_rc, chaining score,
    non_covered_regions, mam_value, mam_solution)] with repeated mam_solutions of a read dropped (align.py:477-496)."""
    instances, owner = [], []
    for r, locs in enumerate(locations):
        rc_seq = None
        for (chr_id, sol, score, is_rc) in locs:
            if is_rc and rc_seq is None:
                rc_seq = reverse_complement(reads[r])
            instances.append((sol, rc_seq if is_rc else reads[r]))
            owner.append((r, chr_id, is_rc, score))
    results = classify_read_with_mams.main_batch(instances, *index, min_acc=min_acc, ctx=ctx) if instances else []
    out = [[] for _ in reads]
    seen = [set() for _ in reads]
    for (r, chr_id, is_rc, score), (non_covered, value, mam_solution) in zip(owner, results):
        if mam_solution in seen[r]:
            continue
        seen[r].add(mam_solution)
        out[r].append((chr_id, is_rc, score, non_covered, value, mam_solution))
    return out


# ---------------------------------------------------------------------------------------------------------------
# align_single, restaged: the per-read loop of /root/reference/modules/align.py:392-560 as batch phases
# ---------------------------------------------------------------------------------------------------------------
# The signature, the queue protocol (batches of [batch_id, [(acc, seq, qual, hits, hits_rc), ...]], the None
# sentinel put back for the other consumers, one list of SAM strings per input batch on output_sam_buffer, the
# 8-entry classification list at the end) and every per-read decision are the reference's.  What changes is the
# order of work: the reads of `args.gpu_batch_reads` queue batches are taken together through
#     seeds -> instances + bounds (host)          -> ONE ultra_chain_nams_batch launch        (:429-453)
#     location selection (host, :469-475)         -> ONE device-planned MAM stage             (:487-489)
#     find_exons / classify_alignment2 (the reference's own modules, unchanged)                (:493-506)
#     base-level alignment of every surviving location in one batch                            (:508)
#     thresholds, primary / secondary, SAM rows (the reference's sam_output, unchanged)        (:510-551)
# sam_output, classify_alignment2 and find_exons are not on the hot path and stay what they are (north_star): they
# are taken from the uLTRA installation this module is dropped into (`modules.*`), or from `args.downstream`.
import itertools
import os
import pickle
import sys
from array import array as _array

INDEX_FILES = ("ref_segment_sequences", "ref_flank_sequences", "splices_to_transcripts", "transcripts_to_splices",
               "all_splice_pairs_annotations", "all_splice_sites_annotations", "parts_to_segments", "segment_to_gene",
               "gene_to_small_segments", "max_intron_chr", "ref_exon_sequences", "chr_to_id", "id_to_chr")
CLASS_OFFSET = {'FSM': 1, 'unaligned': 2, 'NO_SPLICE': 3, 'Insufficient_junction_coverage_unclassified': 4,
                'ISM/NIC_known': 5, 'NIC_novel': 6, 'NNC': 7}


def import_data(args):
    """The 13 index dictionaries a worker reads (align.py:53-110), in the reference's order, from the pickles that
    `uLTRA index` wrote (plain pickle protocol; written by the user's own indexing run)."""
    folder = args.index if getattr(args, "index", "") else args.outfolder
    out = []
    for name in INDEX_FILES:
        with open(os.path.join(folder, name + ".pickle"), "rb") as fh:
            out.append(pickle.load(fh))
    return tuple(out)


def remove_read_polyA_ends(seq, qual, threshold_len, to_len):
    """help_functions.py:41-73: homopolymer runs of A or T longer than threshold_len inside the last
    min(len // 2, 100) bases are cut down to to_len bases (qualities: the first to_len of the run)."""
    window = min(len(seq) // 2, 100)
    head, tail = seq[:-window], seq[-window:]          # window == 0: seq[:-0] is '' and the tail the whole read, as there
    out_seq, out_qual = [head], None
    if qual is not None:
        out_qual = [qual[:-window]]
        tail_qual = qual[-window:]
    pos = 0
    for base, run in itertools.groupby(tail):
        n = sum(1 for _ in run)
        keep = to_len if (n > threshold_len and base in "AT") else n
        out_seq.append(base * keep)
        if out_qual is not None:
            # the reference zips bases with qualities: a quality string shorter than the tail ends the walk early
            out_qual.append(tail_qual[pos:pos + n][:keep])
        pos += n
    if qual is None:
        return "".join(out_seq), None
    if len(tail_qual) < len(tail):                       # zip() semantics of the reference, kept for odd inputs
        return _polyA_zip_form(seq, qual, threshold_len, to_len, window)
    return "".join(out_seq), "".join(out_qual)


def _polyA_zip_form(seq, qual, threshold_len, to_len, window):
    out_seq, out_qual = [seq[:-window]], [qual[:-window]]
    pairs = list(zip(seq[-window:], qual[-window:]))
    for base, grp in itertools.groupby(pairs, key=lambda p: p[0]):
        quals = [q for _, q in grp]
        keep = to_len if (len(quals) > threshold_len and base in "AT") else len(quals)
        out_seq.append(base * keep)
        out_qual.extend(quals[:keep])
    return "".join(out_seq), "".join(out_qual)


_OP_CHARS = "=IDX"


def _script_to_alignment(ops, query, ref):
    """Gapped strings of an edit script (codes 0 '=', 1 'I', 2 'D', 3 'X') the way help_functions.cigar_to_seq
    (:136-179) lays them out: 'I' puts a query base against '-', 'D' a reference base against '-'."""
    q_aln, r_aln, qi, ri = [], [], 0, 0
    for op in ops:
        if op == 1:
            q_aln.append(query[qi]); r_aln.append('-'); qi += 1
        elif op == 2:
            q_aln.append('-'); r_aln.append(ref[ri]); ri += 1
        else:
            q_aln.append(query[qi]); r_aln.append(ref[ri]); qi += 1; ri += 1
    return "".join(q_aln), "".join(r_aln)


def _takes_edlib_route(read_seq, created_ref_seq, mam_sol_exons_length):
    # align.py:341: very long sequences go through edlib's HW mode instead of parasail
    return len(created_ref_seq) > 20000 or len(read_seq) > 20000 or (1000 < len(read_seq) < mam_sol_exons_length / 10)


def get_exact_alignment_batch(jobs, ctx=None):
    """jobs: [(read_seq, created_ref_seq, mam_sol_exons_length)] -> [(read_aln, ref_aln, alignment_score)], what
    align.get_exact_alignment (align.py:338-354) returns for each, with parasail.sg_trace_scan replaced by ONE
    ultra_sg_affine_batch launch and edlib (the > 20 kb route, help_functions.py:181-192) by ultra_myers_hw_batch."""
    import numpy as np
    from . import lib as _lib
    ctx = ctx or _lib.default_context()
    out = [None] * len(jobs)
    routes = [_takes_edlib_route(*j) for j in jobs]
    for want_edlib in (False, True):
        sel = [i for i, r in enumerate(routes) if r == want_edlib]
        if not sel:
            continue
        chunks, size, qo, ql, to, tl = [], 0, [], [], [], []
        for i in sel:
            for seq, offs, lens in ((jobs[i][0], qo, ql), (jobs[i][1], to, tl)):
                b = seq.encode()
                offs.append(size); lens.append(len(b)); chunks.append(b); size += len(b)
        arena = np.frombuffer(b"".join(chunks) or b"\0", np.uint8)
        if not want_edlib:
            res = ctx.sg_affine_batch(arena, qo, ql, to, tl, 2, -2, 3, 1, want_ops=True)
            for k, i in enumerate(sel):
                o = int(res["ops_off"][k])
                ops = res["ops"][o:o + int(res["n_ops"][k])].tolist()
                read_aln, ref_aln = _script_to_alignment(ops, jobs[i][0], jobs[i][1])
                out[i] = (read_aln, ref_aln, int(res["score"][k]))
        else:
            res = ctx.myers_hw_batch(arena, qo, ql, to, tl, [-1] * len(sel), want_ops=True)
            for k, i in enumerate(sel):
                read_seq, ref_seq = jobs[i][0], jobs[i][1]
                lo = int(res["loc_off"][k])
                start, stop = int(res["loc_start"][lo]), int(res["loc_end"][lo])
                o = int(res["ops_off"][k])
                ops = res["ops"][o:o + int(res["n_ops"][k])][::-1].tolist()
                # help_functions.edlib_alignment (:181-192) hands cigar_to_seq ref_seq[start:stop] - one base short of the
                # location - and pads with ref_seq[stop:]: string slicing forgives the missing base, so does this
                piece = ref_seq[start:stop]
                q_aln, r_aln, qi, ri = [], [], 0, 0
                for op in ops:
                    if op == 1:
                        q_aln.append(read_seq[qi:qi + 1]); r_aln.append('-'); qi += 1
                    elif op == 2:
                        q_aln.append('-'); r_aln.append(piece[ri:ri + 1]); ri += 1
                    else:
                        q_aln.append(read_seq[qi:qi + 1]); r_aln.append(piece[ri:ri + 1]); qi += 1; ri += 1
                read_aln = "-" * start + "".join(q_aln) + "-" * (len(ref_seq) - stop - 1)
                ref_aln = ref_seq[:start] + "".join(r_aln) + ref_seq[stop:]
                if len(read_aln) != len(ref_aln):
                    raise AssertionError("edlib route: alignment strings differ in length (help_functions.py:191)")
                match_score = sum(2 for a, b in zip(read_aln, ref_aln) if a == b)
                out[i] = (read_aln, ref_aln, match_score - 2 * int(res["ed"][k]))
    return out


def get_exact_alignment(read_seq, created_ref_seq, mam_sol_exons_length, ctx=None):
    """align.py:338 for one pair (one GPU batch per call: use get_exact_alignment_batch for throughput)."""
    return get_exact_alignment_batch([(read_seq, created_ref_seq, mam_sol_exons_length)], ctx=ctx)[0]


class Downstream(object):
    """The stages behind the MAM stage that stay the reference's own code.  The default takes them from the uLTRA
    installation (`modules.align.find_exons`, `modules.classify_alignment2.main`, `modules.align.get_exact_alignment`,
    `modules.sam_output.main`); tests and the SIRV plumbing run pass their own.  The base-level alignment
    (get_exact_alignment) defaults to this package's GPU batch (get_exact_alignment_batch)."""

    def __init__(self, find_exons=None, classify=None, exact_alignment=None, sam_entry=None, exact_alignment_batch=None):
        if exact_alignment is None and exact_alignment_batch is None:
            exact_alignment_batch = get_exact_alignment_batch      # row f2: base-level alignment on the GPU, one batch
        if None in (find_exons, classify, sam_entry):
            try:
                from modules import align as ref_align
                from modules import classify_alignment2, sam_output
            except ImportError as e:
                raise ImportError("ultra_amd.align.align_single replaces uLTRA's MAM stage only: find_exons, classify_alignment2, "
                                  "get_exact_alignment and sam_output are taken from the uLTRA installation (`modules.*`), which "
                                  "is not importable here (%s); pass args.downstream = ultra_amd.align.Downstream(...) instead" % e)
            find_exons = find_exons or ref_align.find_exons
            classify = classify or classify_alignment2.main
            sam_entry = sam_entry or sam_output.main
        self.find_exons, self.classify, self.sam_entry = find_exons, classify, sam_entry
        self.exact_alignment, self.exact_alignment_batch = exact_alignment, exact_alignment_batch

    def align_all(self, jobs):
        """jobs: [(read_seq, created_ref_seq, mam_sol_exons_length)] -> [(read_aln, ref_aln, alignment_score)]"""
        if self.exact_alignment_batch is not None:
            return self.exact_alignment_batch(jobs)
        return [self.exact_alignment(*j) for j in jobs]


def _pyset_matches_interpreter(so, n_sets=24):
    """ultra_pyset_order (the library's restatement of CPython's set iteration order for int triples) against this
    interpreter's own sets: a few random sets of every table size the planners meet."""
    import ctypes as C
    import random
    rnd = random.Random(20241008)
    for t in range(n_sets):
        n = (2, 3, 5, 8, 13, 21, 40, 100)[t % 8]
        triples = []
        seen = set()
        while len(triples) < n:
            v = (rnd.randrange(1, 30), rnd.randrange(0, 250000000), rnd.randrange(0, 250000000))
            if v not in seen:
                seen.add(v); triples.append(v)
        s = set()
        for v in triples:
            s.add(v)
        flat = (C.c_int64 * (3 * n))(*[x for v in triples for x in v])
        order = (C.c_int32 * n)()
        if so.ultra_pyset_order(flat, n, order) != 0:
            return False
        if [triples[order[k]] for k in range(n)] != list(s):
            return False
    return True


class AlignState(object):
    """What an align_single worker keeps between batches: index dictionaries, packed index on host and GPU, context."""

    def __init__(self, data, ctx=None, device=None):
        from . import index as native_index
        from . import lib as _lib
        (self.ref_segment_sequences, self.ref_flank_sequences, self.splices_to_transcripts, self.transcripts_to_splices,
         self.all_splice_pairs_annotations, self.all_splice_sites_annotations, self.parts_to_segments, self.segment_to_gene,
         self.gene_to_small_segments, self.max_intron_chr, self.ref_exon_sequences, self.chr_to_id, self.id_to_chr) = data
        self.ctx = ctx or _lib.default_context()
        self.notes = []                 # lines for the worker's uLTRA_batch_N.stderr
        self.python = "%d.%d.%d" % sys.version_info[:3]
        # Where the reference's candidate order follows the iteration order of a Python set (hit segments with equal
        # starts), the planners reproduce CPython's tuple hash and set probing (csrc/pyset.h, written against 3.10).  The
        # order uLTRA itself would produce is the running interpreter's: check the restatement against it, here and now.
        if not _pyset_matches_interpreter(_lib.load()):
            raise RuntimeError("ultra_amd: csrc/pyset.h does not reproduce the set iteration order of this interpreter (Python %s); "
                               "the MAM stage would pick other partial segments than uLTRA does under it" % self.python)
        self.native = native_index.NativeIndex(self.ref_segment_sequences, self.ref_flank_sequences, self.parts_to_segments,
                                               self.segment_to_gene, self.gene_to_small_segments)
        dev = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
        try:
            self.dindex = native_index.DeviceIndex(self.native, dev)
        except _lib.UltraError as e:
            if e.code != -4:            # out of memory, wrong device, ...: not something to work around silently
                raise
            self.dindex = None          # index outside the device planner's envelope: batches are planned on the host
            self.notes.append("device planner not usable for this index (%s): batches are planned on the host" % e)
        self._parts = {}
        self.n_quadratic = self.n_nlogn = 0
        from . import front as _front
        # phases 1-4 of align_reads_batch natively (csrc/front.hip); without a device index the batch's MAM stage is
        # planned on the host inside mam_stage() below
        self.front = _front.Front(self.ctx, self.dindex, self.native, self.max_intron_chr)

    def five(self):
        return (self.ref_segment_sequences, self.ref_flank_sequences, self.parts_to_segments, self.segment_to_gene,
                self.gene_to_small_segments)

    def mam_stage(self, instances, min_acc):
        """[(solution, read_seq)] -> [(non_covered_regions, value, mam_solution)]: one device-planned MAM stage."""
        from . import index as native_index
        from . import lib as _lib
        import numpy as np
        if not instances:
            return []
        if self.dindex is not None:
            flat = native_index.flatten_instances(instances, self._parts)
            try:
                self.ctx.mam_stage_nams(self.dindex, *flat)
                self.ctx.mam_run(min_acc)
                res = self.ctx.mam_fetch(want_names=True)
                return classify_read_with_mams.results_from_names(res, np.diff(flat[5]).tolist())
            except _lib.UltraError as e:
                if e.code != -4:        # only ULTRA_E_UNSUPPORTED means "plan this batch on the host"
                    raise
        return classify_read_with_mams.main_batch(instances, *self.five(), min_acc=min_acc, ctx=self.ctx, index=self.native)

    def close(self):
        self.front.close()
        if self.dindex is not None:
            self.dindex.close()
        self.native.close()


def align_reads_batch(records, state, args, downstream):
    """records: [(read_acc, seq, qual, hits, hits_rc)].  Returns ([SAM rows of read 0], [SAM rows of read 1], ...) and
    the additions to the 8-entry classification list, exactly what the reference's loop (align.py:427-551) produces
    for these reads."""
    n = len(records)
    counts = [0] * 8
    # ---- phases 1-4, natively for the whole batch (csrc/front.hip): hit lines -> instances + bounds -> ONE NAM chaining
    #      launch -> location selection -> ONE MAM stage over every (read, location); poly-A trimming and reverse
    #      complements of the reads included.  Python objects are made below for what the later phases touch. ----
    from . import front as _front
    fr = state.front.run(_front.PackedBatch.from_records(records), args.max_intron, args.dropoff, args.max_loc, args.reduce_read_ployA,
                         args.min_acc, with_mam_stage=state.dindex is not None)
    state.n_quadratic += fr.n_quadratic; state.n_nlogn += fr.n_nlogn
    loc_off = fr.loc_off.tolist()
    loc_chr, loc_rc, loc_score = fr.loc_chr.tolist(), fr.loc_rc.tolist(), fr.loc_score.tolist()
    intron_limit = [None if v < 0 else v for v in fr.last_limit.tolist()]
    mods = []
    for r, (_, seq, qual, _, _) in enumerate(records):
        mods.append((fr.read_mod_seq(r), None if qual is None else remove_read_polyA_ends(seq, qual, args.reduce_read_ployA, 1)[1]))
    if state.dindex is not None:
        mam_of = fr.mam
    else:       # an index the device planner cannot hold: the MAM stage of the batch planned on the host
        host = state.mam_stage([(fr.solution(l), fr.loc_read_seq(l)) for l in range(fr.n_loc)], args.min_acc)
        mam_of = host.__getitem__
    locations = [[(loc_chr[l], None, loc_score[l], bool(loc_rc[l])) for l in range(loc_off[r], loc_off[r + 1])] for r in range(n)]
    by_read = [[mam_of(l) for l in range(loc_off[r], loc_off[r + 1])] for r in range(n)]
    rc_cache = {}
    for r in range(n):
        for l in range(loc_off[r], loc_off[r + 1]):
            if loc_rc[l] and r not in rc_cache:
                rc_cache[r] = fr.loc_read_seq(l)
    # ---- phase 5: exon reconstruction + classification per surviving location; alignment jobs collected ----
    pending = []        # per read: [(location fields..., job index)]
    jobs = []
    last_state = []     # per read: (read_seq, read_qual, is_rc) as the reference's loop variables stand after its last iteration
    for r in range(n):
        seq_mod, qual_mod = mods[r]
        read_seq, read_qual, is_rc = None, None, False
        seen, cand = set(), []
        for (chr_id, mem_solution, chaining_score, loc_rc), (non_covered, mam_value, mam_solution) in zip(locations[r], by_read[r]):
            is_rc = loc_rc
            if is_rc:
                read_seq = rc_cache[r]
                read_qual = None if qual_mod is None else qual_mod[::-1]
            else:
                read_seq, read_qual = seq_mod, qual_mod
            if mam_solution in seen:            # several optimal NAM chainings can lead to the same MAM solution (:493)
                continue
            seen.add(mam_solution)
            exons_len = sum(m.y - m.x for m in mam_solution)
            if not mam_value > 0:
                continue
            exons, created_ref_seq, predicted_exons, predicted_splices, covered = downstream.find_exons(
                chr_id, mam_solution, state.ref_exon_sequences, state.ref_segment_sequences, state.ref_flank_sequences,
                state.all_splice_pairs_annotations)
            classification, transcript_id = downstream.classify(chr_id, predicted_splices, state.splices_to_transcripts,
                                                                state.transcripts_to_splices, state.all_splice_pairs_annotations,
                                                                state.all_splice_sites_annotations)
            largest_intron = max([b.x - a.y for a, b in zip(mam_solution[:-1], mam_solution[1:])]) if len(mam_solution) > 1 else 0
            if largest_intron > intron_limit[r] and classification != 'FSM':      # the limit as the chaining loop left it (:446, :505)
                continue
            jobs.append((read_seq, created_ref_seq, exons_len))
            cand.append((len(jobs) - 1, read_seq, chr_id, classification, predicted_exons, transcript_id, is_rc, covered,
                         non_covered, mam_solution))
        pending.append(cand)
        last_state.append((read_seq, read_qual, is_rc))
    # ---- phase 6: base-level alignments, one batch ----
    aligned = downstream.align_all(jobs)
    # ---- phase 7: thresholds, primary / secondary, SAM rows ----
    out = []
    for r in range(n):
        read_acc = records[r][0]
        seq_mod, qual_mod = mods[r]
        rows = []
        if not locations[r]:
            rows.append(downstream.sam_entry(read_acc, seq_mod, qual_mod, '*', 'unaligned', [], '*', '*', '*', False, False, 0))
            out.append(rows)
            continue
        read_seq, read_qual, is_rc = last_state[r]
        read_alignments = []
        for (job, loc_seq, chr_id, classification, predicted_exons, transcript_id, loc_rc, covered, non_covered, mam_solution) in pending[r]:
            read_aln, ref_aln, alignment_score = aligned[job]
            if alignment_score < 2 * args.alignment_threshold * len(loc_seq) and classification != 'FSM':
                continue
            if len(non_covered) >= 3 and max(non_covered[1:-1]) > args.non_covered_cutoff:
                classification = 'Insufficient_junction_coverage_unclassified'
            read_alignments.append((alignment_score, mam_solution[0].x, mam_solution[-1].y, read_acc, chr_id, classification,
                                    predicted_exons, read_aln, ref_aln, transcript_id, loc_rc, covered / float(len(loc_seq))))
        if not read_alignments:
            rows.append(downstream.sam_entry(read_acc, read_seq, read_qual, '*', 'unaligned', [], '*', '*', '*', is_rc, False, 0))
            out.append(rows)
            continue
        ranked = sorted(read_alignments, key=lambda t: (-t[0], t[2] - t[1], t[5]))
        for i, (score, g0, g1, acc, chr_id, classification, predicted_exons, read_aln, ref_aln, transcript_id, loc_rc, coverage) in enumerate(ranked):
            if i == 0:
                map_score = 0 if (len(ranked) > 1 and score == ranked[1][0]) else 60
                counts[0] += coverage
                counts[CLASS_OFFSET[classification]] += 1
            else:
                map_score = 0
            # the reference hands every row the read as its LAST location iteration left it (:550), not the row's own strand
            rows.append(downstream.sam_entry(acc, read_seq, read_qual, state.id_to_chr[chr_id], classification, predicted_exons, read_aln,
                                             ref_aln, transcript_id, loc_rc, i > 0, map_score, aln_score=score))
        out.append(rows)
    return out, counts


def align_reads_isolated(records, state, args, downstream, log=None):
    """align_reads_batch with per-read isolation.  An input outside the library's envelope (ULTRA_E_UNSUPPORTED: more than
    36 distinct bytes in a batch, an instance beyond the chainers' or the MAM builder's LDS limits, a base-level
    alignment beyond the trace budget) fails the WHOLE launch it is part of; the reference aligns any read.  The batch
    is then cut in halves until the offending reads stand alone; those are reported unaligned, with a line in the
    worker's log, and every other read of the batch gets exactly the rows it would have had (results do not depend on
    the grouping: test_align_single_is_independent_of_gpu_batching)."""
    from . import lib as _lib
    try:
        return align_reads_batch(records, state, args, downstream)
    except _lib.UltraError as e:
        if e.code != -4:
            raise
        if len(records) == 1:
            acc, seq, qual = records[0][0], records[0][1], records[0][2]
            if log is not None:
                log.write("ultra_amd: read %s (%d bases) is outside the GPU path's envelope and is reported unaligned: %s\n" % (acc, len(seq), e))
                log.flush()
            seq_mod, qual_mod = remove_read_polyA_ends(seq, qual, args.reduce_read_ployA, 1)
            return [[downstream.sam_entry(acc, seq_mod, qual_mod, '*', 'unaligned', [], '*', '*', '*', False, False, 0)]], [0] * 8
        half = len(records) // 2
        rows_a, counts_a = align_reads_isolated(records[:half], state, args, downstream, log)
        rows_b, counts_b = align_reads_isolated(records[half:], state, args, downstream, log)
        return rows_a + rows_b, [a + b for a, b in zip(counts_a, counts_b)]


def align_single(process_id, input_queue, output_sam_buffer, classification_and_aln_cov, args):
    """Drop-in for modules/align.py:392 (started by pc.py:81).  One worker drives one GPU: LOCAL_RANK, or
    process_id modulo the number of GPUs.  args may carry, beyond the reference's fields: gpu_batch_reads (reads taken
    through the phases together, default 65536), downstream (ultra_amd.align.Downstream), ultra_state (a prepared
    AlignState, used by tests)."""
    from . import lib as _lib
    state = getattr(args, "ultra_state", None)
    own_state = state is None
    if own_state:
        n_dev = max(1, _lib.load().ultra_device_count())
        device = int(os.environ.get("LOCAL_RANK", process_id % n_dev))
        state = AlignState(import_data(args), ctx=_lib.Context(device), device=device)
    downstream = getattr(args, "downstream", None) or Downstream()
    group_reads = int(getattr(args, "gpu_batch_reads", 65536))
    log_path = os.path.join(args.outfolder, "uLTRA_batch_{0}.stderr".format(process_id))
    warning_log_file = open(log_path, "w")
    warning_log_file.write("ultra_amd: GPU worker %d on device %s, GPU_MAX_HW_QUEUES=%s, Python %s\n"
                           % (process_id, os.environ.get("LOCAL_RANK", "?"), os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default 4)"),
                              state.python))
    for note in state.notes:
        warning_log_file.write("ultra_amd: %s\n" % note)
    warning_log_file.flush()
    classification_list = [0, 0, 0, 0, 0, 0, 0, 0]
    processed = 0
    done = False
    while not done:
        group, n_reads = [], 0
        while n_reads < group_reads:
            if group and input_queue.empty():
                break                                   # do not wait for more while reads are at hand
            batch = input_queue.get()
            if batch is None:
                input_queue.put(batch)                  # the signal stays for the other consumers (:419-423)
                done = True
                break
            group.append(batch)
            n_reads += len(batch[1])
        if not group:
            continue
        records = [tuple(b) for batch in group for b in batch[1]]
        rows, counts = align_reads_isolated(records, state, args, downstream, warning_log_file)
        for i in range(8):
            classification_list[i] += counts[i]
        at = 0
        for batch in group:                             # one output list per input batch, as the reference puts them (:553)
            k = len(batch[1])
            output_sam_buffer.put([row for per_read in rows[at:at + k] for row in per_read])
            at += k
        processed += len(records)
    classification_and_aln_cov.put(classification_list)
    print('Process:', classification_list)
    warning_log_file.close()
    print("Number of reads processed by GPU worker {0}: {1}".format(process_id, processed))
    if own_state:
        state.close()
        state.ctx.close()
