"""Generates tests/golden/index_golden.json by RUNNING THE REFERENCE'S OWN
modules/create_augmented_gene.create_graph_from_exon_parts on the bundled SIRV annotation (tests/golden/sirv/, the
data files of /root/reference/test/).

    PYTHONHASHSEED=0 python tests/golden/make_index_golden.py      (needs /root/reference)

What could not be imported: gffutils (not in this image).  The reference only asks its database for three things -
exons ordered by seqid, transcripts ordered by seqid, a transcript's exons ordered by start - so the generator hands
it a duck-typed database over the rows of the GTF.  The row order inside a seqid is the assumption written down in
ultra_amd/indexer.py: (start, end), as gffutils' (seqid, start, end) index returns them; the reference's own
assertion at :440 holds under it and fails under file order.  Exon ids are numbered as gffutils numbers them for GTF
input.  `edlib`, `parasail` (imported by help_functions, never called here) are empty modules.

The parameter sets: the defaults of `uLTRA index` (flank 1000, small exon 200, min segment 25), plus two that push
more intervals through the widening branches (:121-257).
"""
import contextlib
import io
import json
import os
import sys
import types
from array import array

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.modules["edlib"] = types.ModuleType("edlib")
sys.modules["parasail"] = types.ModuleType("parasail")

from modules import create_augmented_gene as ref       # noqa: E402
from ultra_amd import indexer                          # noqa: E402  (only its GTF/FASTA readers)


class Feature(object):
    def __init__(self, fid, seqid, start, stop, attributes):
        self.id, self.seqid, self.start, self.stop, self.end, self.attributes = fid, seqid, start, stop, stop, attributes


class DuckDB(object):
    def __init__(self, exons):
        self.exons = [Feature(e.id, e.seqid, e.start, e.stop, {"gene_id": list(e.gene_ids), "transcript_id": [e.transcript_id]}) for e in exons]
        self.by_tr = {}
        for f in self.exons:
            self.by_tr.setdefault(f.attributes["transcript_id"][0], []).append(f)

    def features_of_type(self, kind, order_by=None):
        if kind == "exon":
            return iter(sorted(self.exons, key=lambda f: (f.seqid, f.start, f.stop)))
        assert kind == "transcript"
        trs = [Feature(tid, fs[0].seqid, min(f.start for f in fs), max(f.stop for f in fs), {}) for tid, fs in self.by_tr.items()]
        return iter(sorted(trs, key=lambda f: f.seqid))

    def children(self, transcript, featuretype=None, order_by=None):
        return iter(sorted(self.by_tr[transcript.id], key=lambda f: f.start))


def triple(key):
    a = array('L'); a.frombytes(key)
    return [int(v) for v in a]


def main():
    d = os.path.join(HERE, "sirv")
    refs = indexer.read_fasta(os.path.join(d, "SIRV_genes.fasta"))
    lengths = {k: len(v) for k, v in refs.items()}
    exons = indexer.read_gtf(os.path.join(d, "SIRV_genes_C_170612a.gtf"))
    out = {"source": "reference modules/create_augmented_gene.create_graph_from_exon_parts over a duck-typed database "
                     "of tests/golden/sirv/SIRV_genes_C_170612a.gtf", "cases": []}
    for flank, small, min_segm in [(1000, 200, 25), (300, 120, 60), (2000, 400, 8)]:
        with contextlib.redirect_stdout(io.StringIO()):
            (segment_to_ref, parts_to_segments, splices_to_transcripts, transcripts_to_splices, pairs, sites, coords,
             segment_to_gene, gene_to_small, flanks, max_intron, exon_ids, chr_to_id, id_to_chr) = \
                ref.create_graph_from_exon_parts(DuckDB(exons), flank, small, min_segm, lengths)
        refs_id = {chr_to_id[k]: v for k, v in refs.items() if k in chr_to_id}
        seqs = ref.get_sequences_from_choordinates(list(coords)[:40], refs_id)
        out["cases"].append({
            "flank_size": flank, "small_exon_threshold": small, "min_segm": min_segm,
            "segment_to_ref": sorted(triple(k) + [int(v)] for k, v in segment_to_ref.items()),
            "parts_to_segments": sorted([triple(k), [int(v) for v in flat]] for k, flat in parts_to_segments.items()),
            "splices_to_transcripts": sorted([int(c), [list(s) for s in sp], sorted(t)] for c, tab in splices_to_transcripts.items() for sp, t in tab.items()),
            "transcripts_to_splices": sorted([int(c), t, [list(s) for s in sp]] for c, tab in transcripts_to_splices.items() for t, sp in tab.items()),
            "all_splice_pairs_annotations": sorted([int(c), list(p), sorted(t)] for c, tab in pairs.items() for p, t in tab.items()),
            "all_splice_sites_annotations": sorted([int(c), sorted(int(x) for x in s)] for c, s in sites.items()),
            "segment_id_to_choordinates": sorted(triple(k) + [int(v[0]), int(v[1])] for k, v in coords.items()),
            "segment_to_gene": sorted([triple(k), sorted(g)] for k, g in segment_to_gene.items()),
            "gene_to_small_segments": sorted([g, [int(v) for v in flat]] for g, flat in gene_to_small.items()),
            "flanks": sorted(triple(k) for k in flanks),
            "max_intron_chr": sorted([int(c), int(v)] for c, v in max_intron.items()),
            "exon_ids": sorted(triple(k) for k in exon_ids),
            "chr_to_id": sorted([k, int(v)] for k, v in chr_to_id.items()),
            "segment_sequences_sample": sorted(triple(k) + [v] for k, v in seqs.items()),
        })
    with open(os.path.join(HERE, "index_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote index_golden.json:", [(len(c["parts_to_segments"]), len(c["segment_to_ref"]), len(c["flanks"])) for c in out["cases"]])


if __name__ == "__main__":
    main()
