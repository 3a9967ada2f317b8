"""ultra_amd.indexer against the reference's create_graph_from_exon_parts on the bundled SIRV annotation
(tests/golden/index_golden.json, made by tests/golden/make_index_golden.py)."""
import json
import os
from array import array

import pytest

from ultra_amd import indexer

HERE = os.path.dirname(os.path.abspath(__file__))
SIRV = os.path.join(HERE, "golden", "sirv")


def triple(key):
    a = array('L'); a.frombytes(key)
    return [int(v) for v in a]


def triples_of(flat):
    return sorted(tuple(flat[i:i + 3]) for i in range(0, len(flat), 3))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "index_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def inputs():
    refs = indexer.read_fasta(os.path.join(SIRV, "SIRV_genes.fasta"))
    return refs, indexer.read_gtf(os.path.join(SIRV, "SIRV_genes_C_170612a.gtf"))


def test_gtf_reader(inputs):
    refs, exons = inputs
    assert len(exons) == 339 and exons[0].id == "exon_1" and exons[0].gene_ids == ["SIRV1"]
    assert (exons[0].seqid, exons[0].start, exons[0].stop, exons[0].transcript_id) == ("SIRV1", 1001, 1484, "SIRV101")
    assert sorted(refs) == ["SIRV%d" % i for i in range(1, 8)]


@pytest.mark.parametrize("case_no", [0, 1, 2])
def test_annotation_matches_the_reference(golden, inputs, case_no):
    refs, exons = inputs
    c = golden["cases"][case_no]
    ann = indexer.build_annotation(exons, c["flank_size"], c["small_exon_threshold"], c["min_segm"], {k: len(v) for k, v in refs.items()})
    assert sorted([k, v] for k, v in ann["chr_to_id"].items()) == c["chr_to_id"]
    assert {v: k for k, v in ann["chr_to_id"].items()} == ann["id_to_chr"]
    assert sorted(triple(k) for k in ann["flank_choordinates"]) == c["flanks"]
    assert sorted(triple(k) for k in ann["exon_choordinates_to_id"]) == c["exon_ids"]
    assert sorted(triple(k) + [v] for k, v in ann["segment_to_ref"].items()) == c["segment_to_ref"]
    assert sorted(triple(k) + list(v) for k, v in ann["segment_id_to_choordinates"].items()) == c["segment_id_to_choordinates"]
    assert sorted([triple(k), sorted(g)] for k, g in ann["segment_to_gene"].items()) == c["segment_to_gene"]
    # the arrays are walked as sets of (chr, start, stop) downstream (classify_read_with_mams.py:180-300); their
    # order inside a part follows the iteration order of sets of exon ids in the reference, so compare as multisets
    mine = {tuple(triple(k)): triples_of(list(v)) for k, v in ann["parts_to_segments"].items()}
    ref = {tuple(k): triples_of(v) for k, v in c["parts_to_segments"]}
    assert mine == ref
    mine = {g: triples_of(list(v)) for g, v in ann["gene_to_small_segments"].items()}
    assert mine == {g: triples_of(v) for g, v in c["gene_to_small_segments"]}
    assert sorted([int(k), int(v)] for k, v in ann["max_intron_chr"].items()) == c["max_intron_chr"]
    assert sorted([int(ch), sorted(s)] for ch, s in ann["all_splice_sites_annotations"].items()) == c["all_splice_sites_annotations"]
    assert sorted([int(ch), list(p), sorted(t)] for ch, tab in ann["all_splice_pairs_annotations"].items() for p, t in tab.items()) == c["all_splice_pairs_annotations"]
    assert sorted([int(ch), [list(s) for s in sp], sorted(t)] for ch, tab in ann["splices_to_transcripts"].items() for sp, t in tab.items()) == c["splices_to_transcripts"]
    assert sorted([int(ch), t, [list(s) for s in sp]] for ch, tab in ann["transcripts_to_splices"].items() for t, sp in tab.items()) == c["transcripts_to_splices"]
    indexer.add_sequences(ann, refs)
    for ch, a, b, seq in c["segment_sequences_sample"]:
        assert ann["ref_segment_sequences"][array("L", [ch, a, b]).tobytes()] == seq
    assert set(ann["ref_part_sequences"]) == set(ann["parts_to_segments"]) | set(ann["flank_choordinates"])


def test_index_folder_round_trip(tmp_path, inputs):
    import pickle
    ann = indexer.build_index(os.path.join(SIRV, "SIRV_genes_C_170612a.gtf"), os.path.join(SIRV, "SIRV_genes.fasta"), str(tmp_path))
    from ultra_amd.align import INDEX_FILES
    for name in INDEX_FILES:
        with open(os.path.join(str(tmp_path), name + ".pickle"), "rb") as f:
            got = pickle.load(f)
        assert len(got) == len(ann[name])


def test_seeder_hits_are_maximal_exact_matches(inputs):
    import numpy as np
    from ultra_amd import seeder
    refs, exons = inputs
    ann = indexer.add_sequences(indexer.build_annotation(exons, refs_lengths={k: len(v) for k, v in refs.items()}), refs)
    index = seeder.MemIndex(ann["ref_part_sequences"], min_len=14)
    by_name = dict(zip(index.names, index.seqs))
    rng = np.random.RandomState(5)
    keys = sorted(ann["parts_to_segments"])
    n = 0
    for t in range(12):
        part = ann["ref_part_sequences"][keys[int(rng.randint(len(keys)))]].upper()
        a = int(rng.randint(0, max(1, len(part) - 80)))
        read = list(part[a:a + 300])
        for p in rng.randint(0, len(read), len(read) // 25):
            read[p] = "ACGT"[("ACGT".index(read[p]) + 1) % 4] if read[p] in "ACGT" else "A"
        read = "".join(read)
        for name, rpos, qpos, ln in index.mems(read):
            ref = by_name[name]
            assert ln >= 14 and ref[rpos - 1:rpos - 1 + ln] == read[qpos - 1:qpos - 1 + ln]
            assert rpos == 1 or qpos == 1 or ref[rpos - 2] != read[qpos - 2]
            assert rpos - 1 + ln == len(ref) or qpos - 1 + ln == len(read) or ref[rpos - 1 + ln] != read[qpos - 1 + ln]
            n += 1
    assert n > 20
    assert seeder.reverse_complement("ACGTTn") == "nAACGT"
