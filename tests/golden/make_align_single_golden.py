"""Generates tests/golden/align_single_golden.json by RUNNING THE REFERENCE'S OWN modules/align.align_single (the
consumer loop, /root/reference/modules/align.py:392-560) in the build container, on the bundled SIRV annotation.

    PYTHONHASHSEED=0 python tests/golden/make_align_single_golden.py      (needs /root/reference)

What runs for real: align_single and everything it calls inside the reference - get_mems_from_input,
annotate_guaranteed_optimal_bound, both NAM chainers, classify_read_with_mams.main, find_exons,
classify_alignment2.main, get_exact_alignment, sam_output.main.  What could not be imported and is bound to a
stand-in (none of these wheels exists in this image):
  edlib.align                  -> oracle/cport.hw_align   (as in make_classify_golden.py)
  parasail.sg_trace_scan_16    -> oracle/cport.sg_affine  (the affine-gap restatement; parity with parasail itself
                                  is unpinned - the vectors pin what the reference does AROUND the call)
  pysam, gffutils              -> empty modules (imported at module level, not used by this path)
Inputs: the index dictionaries are ultra_amd.indexer's (equal to the reference's construction, tests/test_indexer.py),
written as the pickles import_data reads; reads are the 4 bundled reads plus reads simulated from the annotated SIRV
transcripts (errors, both strands, poly-A tails, some with qualities, some junk); seeds are ultra_amd.seeder's.

The fixture is data only: the reads and hit lines that went in, the SAM rows and the classification list that came
out, and a transcript of every call into the four downstream stages (arguments digest -> return value), which the GPU
test replays so that it needs neither the reference nor its dependencies.
"""
import contextlib
import io
import json
import os
import queue
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import cport                                   # noqa: E402
from tests import helpers                                  # noqa: E402
from tests.helpers import digest, jsonable, mam_key       # noqa: E402
from ultra_amd import indexer, pc, seed_wrapper, seeder    # noqa: E402


def _edlib_align(query, target, task="distance", mode="NW", k=-1, additionalEqualities=None):
    assert mode == "HW" and task == "path"
    r = cport.hw_align(query, target, int(k))
    return {"editDistance": r["editDistance"], "alphabetLength": r["alphabetLength"], "locations": r["locations"], "cigar": r["cigar"]}


class _Cigar(object):
    def __init__(self, text):
        self.decode = text.encode()


class _Result(object):
    def __init__(self, r):
        self.saturated, self.score, self.cigar = False, r["score"], _Cigar(r["cigar"])
        self.end_query, self.end_ref = r["end_query"], r["end_ref"]


def _sg_trace(s1, s2, gap_open, gap_extend, matrix):
    return _Result(cport.sg_affine(s1, s2, matrix[1], matrix[2], gap_open, gap_extend))


edlib_mod = types.ModuleType("edlib"); edlib_mod.align = _edlib_align
parasail_mod = types.ModuleType("parasail")
parasail_mod.matrix_create = lambda alphabet, match, mismatch: (alphabet, match, mismatch)
parasail_mod.sg_trace_scan_16 = _sg_trace
parasail_mod.sg_trace_scan_32 = _sg_trace
sys.modules.update({"edlib": edlib_mod, "parasail": parasail_mod, "pysam": types.ModuleType("pysam"), "gffutils": types.ModuleType("gffutils")})

from modules import align as ref_align                     # noqa: E402
from modules import classify_alignment2, sam_output        # noqa: E402


def main():
    d = os.path.join(HERE, "sirv")
    gtf, fasta = os.path.join(d, "SIRV_genes_C_170612a.gtf"), os.path.join(d, "SIRV_genes.fasta")
    tmp = tempfile.mkdtemp()
    ann = indexer.build_index(gtf, fasta, tmp)
    refs = indexer.read_fasta(fasta)
    rng = np.random.RandomState(20240611)
    with open(os.path.join(d, "reads.fa")) as fh:
        reads = [(acc, seq, qual) for acc, (seq, qual) in pc.readfq(fh)]
    reads += helpers.simulate_transcript_reads(indexer.read_gtf(gtf), refs, rng, 44)
    seeds = os.path.join(tmp, "seeds.txt.gz")
    seeder.write_seed_file(seeds, [(a, s) for a, s, _ in reads], ann["ref_part_sequences"], min_len=15)
    records = [[acc, seq, qual, hits, hits_rc] for (acc, seq, qual), (_, hits, _, hits_rc) in zip(reads, seed_wrapper.read_seeds(seeds))]

    calls = {"find_exons": {}, "classify": {}, "exact_alignment": {}, "sam_entry": {}}

    def record(name, fn, key_of):
        def wrapped(*a, **kw):
            out = fn(*a, **kw)
            calls[name][digest(key_of(a, kw))] = jsonable(out)
            return out
        return wrapped

    ref_align.find_exons = record("find_exons", ref_align.find_exons, lambda a, kw: [a[0], mam_key(a[1])])
    classify_alignment2.main = record("classify", classify_alignment2.main, lambda a, kw: [a[0], a[1]])
    ref_align.get_exact_alignment = record("exact_alignment", ref_align.get_exact_alignment, lambda a, kw: list(a))
    sam_output.main = record("sam_entry", sam_output.main, lambda a, kw: [list(a), kw])

    out = {"source": "reference modules/align.align_single run in the build container; edlib -> oracle hw_align, parasail -> oracle sg_affine",
           "index": {"flank_size": 1000, "small_exon_threshold": 200, "min_segm": 25}, "records": records, "runs": []}
    for name, kw in (("ont", dict(min_acc=0.6, reduce_read_ployA=20, alignment_threshold=0.5, non_covered_cutoff=15, dropoff=0.95, max_loc=5, max_intron=1200000)),
                     ("strict", dict(min_acc=0.8, reduce_read_ployA=8, alignment_threshold=0.7, non_covered_cutoff=5, dropoff=0.8, max_loc=3, max_intron=3000))):
        args = types.SimpleNamespace(outfolder=tmp, index=tmp, **kw)
        qin, qout, qcnt = queue.Queue(), queue.Queue(), queue.Queue()
        for b in range(0, len(records), 20):
            qin.put([b // 20 + 1, records[b:b + 20]])
        qin.put(None)
        with contextlib.redirect_stdout(io.StringIO()):
            ref_align.align_single(0, qin, qout, qcnt, args)
        rows = []
        while not qout.empty():
            rows.append(qout.get())
        out["runs"].append({"name": name, "args": kw, "rows": rows, "classification_list": jsonable(qcnt.get())})
    out["calls"] = calls
    path = os.path.join(HERE, "align_single_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in calls.items()},
          [(r["name"], sum(len(b) for b in r["rows"]), r["classification_list"]) for r in out["runs"]])


if __name__ == "__main__":
    main()
