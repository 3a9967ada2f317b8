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


def chain_reads_batch(seed_records, max_intron_chr, max_global_intron, dropoff, max_loc, ctx=None):
    """seed_records: [(hits, hits_rc)] per read (raw hit lines, seed_wrapper.read_seeds).  Returns per read the
    locations the reference would hand to the MAM stage, in its order: [(chr_id, mem_solution, score, is_rc)]
    (align.py:429-475, before the duplicate-MAM-solution filter of :493 which needs the MAM results)."""
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
        all_chainings, best = [], 0
        for k, ((chr_id, _), (bound, is_rc, _)) in enumerate(order):
            if bound < best * dropoff:
                break
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
