"""Annotation -> index dictionaries, without gffutils (BASELINE.json configs[0]: the bundled SIRV plumbing run).

`uLTRA index` (/root/reference/uLTRA:47-172) builds, from a GTF and a genome FASTA, the dictionaries the aligner
reads: exon PARTS (maximal runs of exons at most 20 bp apart), canonical SEGMENTS inside a part (the intervals between
consecutive exon boundaries, short ones widened or replaced by whole exons), FLANKS between and around parts, and the
splice-site tables.  The reference walks a gffutils database (/root/reference/modules/create_augmented_gene.py:278-539,
segments :63-263, sequences :547-560); gffutils is not part of this image, and indexing is off the hot path and stays
the reference's job in a real installation.  This module exists so that the plumbing configuration can run end to
end here: a small GTF reader and the same construction on plain records.  It is pinned by
tests/golden/index_golden.json, produced by running the reference's create_graph_from_exon_parts on the bundled SIRV
annotation through a duck-typed database (tests/golden/make_index_golden.py).

Order of the records: the reference iterates `db.features_of_type('exon', order_by='seqid')` and asserts that no exon
starts before the part it extends (:440); the bundled annotation is grouped by transcript, so that only holds because
the database hands the rows over along its (seqid, start, end) index.  That is the order assumed here, ties in file
order (ties only permute exon ids inside sets).  Exon ids are gffutils' automatic ones for GTF input ("exon_<n>", n counting the
exon rows of the file from 1).
"""
import pickle
import os
from array import array
from collections import defaultdict, namedtuple

Exon = namedtuple("Exon", ["id", "seqid", "start", "stop", "gene_ids", "transcript_id"])   # start 1-based, stop inclusive (GTF)


def _attributes(field):
    out = {}
    for item in field.strip().split(";"):
        item = item.strip()
        if not item:
            continue
        key, _, value = item.partition(" ")
        out.setdefault(key, []).append(value.strip().strip('"'))
    return out


def read_gtf(path):
    """Exon rows of a GTF as Exon records in file order."""
    exons, n = [], 0
    with open(path) as fh:
        for line in fh:
            if not line.strip() or line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if len(f) < 9 or f[2] != "exon":
                continue
            n += 1
            att = _attributes(f[8])
            exons.append(Exon("exon_%d" % n, f[0], int(f[3]), int(f[4]), sorted(att.get("gene_id", [])),
                              (att.get("transcript_id") or [""])[0]))
    return exons


def read_fasta(path):
    refs, name, chunks = {}, None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    refs[name] = "".join(chunks)
                fields = line[1:].split()
                name, chunks = (fields[0] if fields else ""), []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        refs[name] = "".join(chunks)
    return refs


def _key(chr_id, a, b):
    return array("L", [chr_id, a, b]).tobytes()


def _new_array():       # module-level default factories: the tables keep their default behaviour through pickle
    return array("L")


def _dd_set():
    return defaultdict(set)


class _Part(object):
    __slots__ = ("chr_id", "number", "start", "stop", "positions", "genes", "starts_at", "ends_at")

    def __init__(self, chr_id, number, start, stop):
        self.chr_id, self.number, self.start, self.stop = chr_id, number, start, stop
        self.positions, self.genes = set(), set()
        self.starts_at, self.ends_at = defaultdict(set), defaultdict(set)     # boundary -> exon ids (pos_to_exon_ids)


def _segments_of_parts(parts, exon_span, small_threshold, min_size):
    """get_canonical_segments (:63-263).  Returns parts_to_segments, segment_to_gene, segment coordinates,
    gene_to_small_segments.  Kept on purpose: a part's segments all share ONE gene set object; the forward widening
    adds its segment every time it is reached (its membership test compares a tuple with byte keys, :168); an exon
    taken over as a segment is booked as a small segment of whichever gene a small-segment loop visited last (:251)."""
    parts_to_segments = defaultdict(_new_array)
    gene_to_small = defaultdict(_new_array)
    segment_to_gene, coords = {}, {}
    last_gene = [None]      # the reference's loop variable `gene_id`, which outlives its loops

    def add(part, a, b, small_for_all_genes=True):
        name = _key(part.chr_id, a, b)
        coords[name] = (a, b)
        parts_to_segments[_key(part.chr_id, part.start, part.stop)].extend([part.chr_id, a, b])
        segment_to_gene[name] = part.genes
        if b - a <= small_threshold:
            if small_for_all_genes:
                for g in part.genes:
                    gene_to_small[g].extend([part.chr_id, a, b])
                    last_gene[0] = g
            else:
                if last_gene[0] is None:
                    raise NameError("name 'gene_id' is not defined")      # what the reference would raise here (:251)
                gene_to_small[last_gene[0]].extend([part.chr_id, a, b])

    for part in parts:
        cuts = sorted(part.positions)
        spans = list(zip(cuts[:-1], cuts[1:]))
        open_exons = set()
        for i, (p1, p2) in enumerate(spans):
            open_exons |= part.starts_at[p1]
            open_exons -= part.ends_at[p1]
            if p2 - p1 >= min_size:
                add(part, p1, p2)
                open_exons -= part.ends_at[p2]
                continue
            # too short on its own: widen to the left ...
            for k in range(1, i + 1):
                left = spans[i - k][0]
                if p2 - left >= min_size:
                    if _key(part.chr_id, left, p2) not in coords:
                        add(part, left, p2)
                    break
            # ... and to the right
            for k in range(1, len(spans) - i):
                right = spans[i + k][1]
                if right - p1 >= min_size:
                    add(part, p1, right)
                    break
            if len(spans) == 1:
                add(part, p1, p2)
            # and the exons that begin or end here (or, failing that, the ones open across it) become segments themselves
            spanning = set(part.starts_at[p1]) | set(part.ends_at[p2])
            if not spanning:
                spanning = open_exons
            for e_id in sorted(spanning):
                e_start, e_stop = exon_span[e_id]
                if _key(part.chr_id, e_start, e_stop) not in coords:
                    add(part, e_start, e_stop, small_for_all_genes=False)
            open_exons -= part.ends_at[p2]
    return parts_to_segments, segment_to_gene, coords, gene_to_small


def build_annotation(exons, flank_size=1000, small_exon_threshold=200, min_segm=25, refs_lengths=None):
    """create_graph_from_exon_parts (:278-539) on Exon records.  Returns a dict named like the reference's pickles."""
    refs_lengths = refs_lengths or {}
    if not exons:
        raise ValueError("no exon record in the annotation (is column 3 'exon'?)")
    order = sorted(range(len(exons)), key=lambda i: (exons[i].seqid, exons[i].start, exons[i].stop))
    chr_to_id, id_to_chr = {}, {}
    exon_span, exon_ids, flanks, parts = {}, set(), set(), []
    cur, cur_name, cur_genes = None, None, set()

    def chr_id_of(name):
        if name not in chr_to_id:
            chr_to_id[name] = len(chr_to_id) + 1
            id_to_chr[chr_to_id[name]] = name
        return chr_to_id[name]

    def open_part(chr_id, e, register):
        part = _Part(chr_id, len(parts), e.start - 1, e.stop)
        part.positions.update((e.start - 1, e.stop))
        part.genes.update(e.gene_ids)
        if register:                      # the very first exon of the annotation is never booked by boundary (:329-350)
            part.starts_at[e.start - 1].add(e.id)
            part.ends_at[e.stop].add(e.id)
        parts.append(part)
        return part

    def first_flank(chr_id, e):
        if e.start - 1 > 0:
            flanks.add(_key(chr_id, max(0, e.start - 2 * flank_size), e.start - 1))

    def last_flank(chr_id, name, stop):
        length = refs_lengths.get(name, stop + 2 * flank_size)
        if stop < length:
            flanks.add(_key(chr_id, max(0, stop), min(length, stop + 2 * flank_size)))

    for n, i in enumerate(order):
        e = exons[i]
        chr_id = chr_id_of(e.seqid)
        exon_span[e.id] = (e.start - 1, e.stop)
        exon_ids.add(_key(chr_id, e.start - 1, e.stop))
        if n == 0:
            cur, cur_name, cur_genes = open_part(chr_id, e, register=False), e.seqid, set(e.gene_ids)
            first_flank(chr_id, e)
            continue
        if chr_id != cur.chr_id:
            last_flank(cur.chr_id, cur_name, cur.stop)
            first_flank(chr_id, e)
            cur, cur_name, cur_genes = open_part(chr_id, e, register=True), e.seqid, set(e.gene_ids)
        elif e.start - 1 > cur.stop + 20:
            reach = 2 * flank_size if set(e.gene_ids).isdisjoint(cur_genes) else flank_size
            if e.start - cur.stop > 2 * reach:
                length = refs_lengths.get(e.seqid, cur.stop + reach)
                flanks.add(_key(chr_id, max(0, cur.stop), min(length, cur.stop + reach)))
                flanks.add(_key(chr_id, max(0, e.start - reach), e.start - 1))
            else:                           # the whole gap
                flanks.add(_key(chr_id, max(0, cur.stop), e.start - 1))
            cur, cur_genes = open_part(chr_id, e, register=True), set(e.gene_ids)
        else:
            cur.stop = max(cur.stop, e.stop)
            cur_genes.update(e.gene_ids)
            cur.positions.update((e.start - 1, e.stop))
            cur.genes.update(e.gene_ids)
            cur.starts_at[e.start - 1].add(e.id)
            cur.ends_at[e.stop].add(e.id)
        if cur.start > e.start - 1:
            raise AssertionError("exons of %s are not sorted by start" % e.seqid)
    last_flank(cur.chr_id, cur_name, cur.stop)

    parts_to_segments, segment_to_gene, coords, gene_to_small = _segments_of_parts(parts, exon_span, small_exon_threshold, min_segm)

    # ---- splice tables from the transcripts (:481-527) ----
    by_transcript = defaultdict(list)
    for e in exons:
        by_transcript[(e.seqid, e.transcript_id)].append((e.start - 1, e.stop))
    splices_to_transcripts = defaultdict(_dd_set)
    pair_annotations = defaultdict(_dd_set)
    site_annotations = defaultdict(set)
    max_intron_chr = defaultdict(int)
    for (seqid, tid), spans in by_transcript.items():
        chr_id = chr_to_id[seqid]
        spans.sort(key=lambda s: s[0])
        introns = [(a[1], b[0]) for a, b in zip(spans[:-1], spans[1:])]
        for i1, i2 in introns:
            if i2 - i1 > max_intron_chr[chr_id]:
                max_intron_chr[chr_id] = i2 - i1
            pair_annotations[chr_id][(i1, i2)].add(tid)
            site_annotations[chr_id].update((i1, i2))
        splices_to_transcripts[chr_id][tuple(introns)].add(tid)
        site_annotations[chr_id].update((spans[0][0], spans[-1][1]))
    transcripts_to_splices = defaultdict(dict)
    for chr_id, table in splices_to_transcripts.items():
        for introns, tids in table.items():
            for tid in tids:
                transcripts_to_splices[chr_id][tid] = introns
    return {
        "segment_to_ref": {k: _chr_of(k) for k in coords},
        "parts_to_segments": parts_to_segments, "splices_to_transcripts": splices_to_transcripts,
        "transcripts_to_splices": transcripts_to_splices, "all_splice_pairs_annotations": pair_annotations,
        "all_splice_sites_annotations": site_annotations, "segment_id_to_choordinates": coords,
        "segment_to_gene": segment_to_gene, "gene_to_small_segments": gene_to_small, "flank_choordinates": flanks,
        "max_intron_chr": max_intron_chr, "exon_choordinates_to_id": exon_ids, "chr_to_id": chr_to_id, "id_to_chr": id_to_chr,
    }


def _chr_of(key):
    a = array("L")
    a.frombytes(key)
    return a[0]


def _slices(keys, refs_by_id):
    out = {}
    for key in keys:
        a = array("L")
        a.frombytes(key)
        if a[0] in refs_by_id:
            out[key] = refs_by_id[a[0]][a[1]:a[2]]      # (:547-560)
    return out


def add_sequences(ann, refs):
    """prep_seqs (uLTRA:109-172): the sequence dictionaries, keyed like the coordinate tables."""
    by_id = {ann["chr_to_id"][name]: seq for name, seq in refs.items() if name in ann["chr_to_id"]}
    ann["ref_flank_sequences"] = _slices(ann["flank_choordinates"], by_id)
    parts = _slices(ann["parts_to_segments"], by_id)
    parts.update(ann["ref_flank_sequences"])
    ann["ref_part_sequences"] = parts                   # what the seeder indexes: exon parts and flanks (uLTRA:339-344)
    ann["ref_segment_sequences"] = _slices(ann["segment_id_to_choordinates"], by_id)
    ann["ref_exon_sequences"] = _slices(ann["exon_choordinates_to_id"], by_id)
    ann["refs_lengths"] = {name: len(seq) for name, seq in refs.items()}
    return ann


def write_index(folder, ann):
    """The index folder `uLTRA align` reads (uLTRA:86-99, :168-172): one pickle per dictionary, same file names."""
    os.makedirs(folder, exist_ok=True)
    for name, value in ann.items():
        with open(os.path.join(folder, name + ".pickle"), "wb") as fh:
            pickle.dump(value, fh, pickle.HIGHEST_PROTOCOL)


def build_index(gtf, fasta, folder=None, flank_size=1000, small_exon_threshold=200, min_segm=25):
    refs = read_fasta(fasta)
    ann = build_annotation(read_gtf(gtf), flank_size, small_exon_threshold, min_segm, {k: len(v) for k, v in refs.items()})
    add_sequences(ann, refs)
    if folder:
        write_index(folder, ann)
    return ann


def align_data(ann):
    """The 13 dictionaries of ultra_amd.align.import_data, in its order."""
    from .align import INDEX_FILES
    return tuple(ann[name] for name in INDEX_FILES)
