"""Generates tests/golden/classify_golden.json by RUNNING THE REFERENCE'S OWN
modules/classify_read_with_mams.py in the build container.

    PYTHONHASHSEED=0 python tests/golden/make_classify_golden.py      (needs /root/reference)

The reference module imports the third-party wheels `edlib` and (through help_functions) `parasail`,
neither of which exists in this image.  To execute the reference's own logic
around that call - edlib_alignment/cigar_to_accuracy (:71-119), add_segment_to_mam (:303-402),
candidate enumeration (:180-300) and main (:406-594) incl. both chainers - `edlib.align` is bound to
this repository's CPU oracle (oracle/cport.hw_align; see oracle/edlib_port.c for what is and is not
pinned about it) and `parasail` to an empty module (never called on this path).  What these vectors
pin is therefore everything the reference does AROUND the edlib call, given identical edlib results.

A fixture is data only: the index dictionaries, reads and NAM chains that went in, and the
(non_covered_regions, value, mam_solution) triples that came out.
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

from oracle import cport                       # noqa: E402
from ultra_amd import synth                    # noqa: E402


def _edlib_align(query, target, task="distance", mode="NW", k=-1, additionalEqualities=None):
    assert mode == "HW" and task == "path"
    r = cport.hw_align(query, target, int(k))   # the Cython binding truncates a float k (SURVEY 8c gamma)
    return {"editDistance": r["editDistance"], "alphabetLength": r["alphabetLength"],
            "locations": r["locations"], "cigar": r["cigar"]}


edlib_mod = types.ModuleType("edlib")
edlib_mod.align = _edlib_align
sys.modules["edlib"] = edlib_mod
sys.modules["parasail"] = types.ModuleType("parasail")

from modules import classify_read_with_mams as ref   # noqa: E402


def triple(key):
    a = array('L'); a.frombytes(key)
    return [int(v) for v in a]


def dump_world(w):
    return {
        "segments": [triple(k) + [v] for k, v in w.ref_segment_sequences.items()],
        "flanks": [triple(k) + [v] for k, v in w.ref_flank_sequences.items()],
        "parts_to_segments": [[triple(k), [int(v) for v in flat]] for k, flat in w.parts_to_segments.items()],
        "segment_to_gene": [[triple(k), sorted(g)] for k, g in w.segment_to_gene.items()],
        "gene_to_small_segments": [[g, [int(v) for v in flat]] for g, flat in w.gene_to_small_segments.items()],
    }


def mam_rows(sol):
    rows = []
    for m in sol:
        rows.append([int(m.x), int(m.y), int(m.c), int(m.d), float(m.val).hex(), int(m.j),
                     int(m.min_segment_length), str(m.mam_id), int(m.ref_chr_id)])
    return rows


def main():
    out = {"source": "reference modules/classify_read_with_mams.py run in the build container, edlib.align bound to "
                     "oracle/cport.hw_align", "worlds": []}
    plan = [  # (world seed, genes, kwargs, [(preset, n, seed)])
        (101, 14, {}, [("ont_cdna", 40, 1), ("isoseq", 12, 2), ("ont_drna", 30, 3)]),
        (102, 6, {"small_exon_frac": 0.35, "split_frac": 0.6}, [("ont_cdna", 40, 4), ("ont_drna", 20, 5)]),
    ]
    for wseed, genes, kw, groups in plan:
        w = synth.make_world(seed=wseed, n_genes=genes, **kw)
        idx = w.index()
        rec = {"world": dump_world(w), "instances": []}
        for preset, n, seed in groups:
            min_acc = synth.PRESETS[preset]["min_acc"]
            for read, sol, _ in synth.make_instances(w, n, seed=seed, preset=preset):
                with contextlib.redirect_stdout(io.StringIO()):
                    gaps, value, mams = ref.main(sol, *idx, read, None, min_acc)
                rec["instances"].append({
                    "preset": preset, "min_acc": min_acc, "read": read,
                    "solution": [[int(m.x), int(m.y), int(m.c), int(m.d), int(m.val), int(m.j), m.exon_part_id] for m in sol],
                    "non_covered_regions": [int(g) for g in gaps], "value": float(value).hex(),
                    "mam_solution": mam_rows(mams)})
        out["worlds"].append(rec)

    # G3: add_segment_to_mam alone, all three call shapes (:303-402), incl. low-complexity repeats
    import numpy as np
    from tests import helpers
    rng = np.random.RandomState(33)
    seg_cases = []
    for i in range(160):
        shape = i % 4
        alpha = "ACGT" if i % 5 else "AC"
        read = helpers.rand_seq(rng, int(rng.randint(60, 500)), alpha)
        if shape == 0:      # approximate segment-in-read
            a = int(rng.randint(0, len(read) - 30)); ln = int(rng.randint(9, min(150, len(read) - a)))
            seq = helpers.mutate(rng, read[a:a + ln], float(rng.choice([0.0, 0.05, 0.15, 0.3]))) or "ACGTACGTAC"
            e0 = int(rng.randint(1000, 2000)); e1 = e0 + len(seq)
        elif shape == 1:    # 5-8 bp exact
            ln = int(rng.randint(5, 9)); a = int(rng.randint(0, len(read) - ln))
            seq = read[a:a + ln] if rng.rand() < 0.7 else helpers.rand_seq(rng, ln, alpha)
            e0 = 500; e1 = e0 + ln
        elif shape == 2:    # read contained in a long segment
            pre, post = helpers.rand_seq(rng, int(rng.randint(0, 200))), helpers.rand_seq(rng, int(rng.randint(0, 200)))
            seq = pre + helpers.mutate(rng, read, float(rng.choice([0.0, 0.05, 0.12]))) + post
            e0 = 7000; e1 = e0 + len(seq)
        else:               # partial segment: full coordinates wider than the sequence handed in
            a = int(rng.randint(0, len(read) - 30)); ln = int(rng.randint(6, min(90, len(read) - a)))
            seq = helpers.mutate(rng, read[a:a + ln], 0.05) or "ACGTAC"
            e0 = 100; e1 = e0 + len(seq) + int(rng.randint(0, 400))
        label = ['_full_segment', '_partial_segment_start', '_partial_segment_end', '_full_flank'][int(rng.randint(0, 4))]
        min_acc = float(rng.choice([0.6, 0.8]))
        inst = []
        with contextlib.redirect_stdout(io.StringIO()):
            ref.add_segment_to_mam(read, 3, seq, e0, e1, "id%d" % i, inst, min_acc, label)
        seg_cases.append({"read": read, "seq": seq, "e_start": e0, "e_stop": e1, "segm_id": "id%d" % i,
                          "min_acc": min_acc, "label": label, "mams": mam_rows(inst)})
    out["add_segment_to_mam"] = seg_cases

    # cigar_to_accuracy (:71-94) on oracle cigars: pins the closed form (L - ED + I) / (L + I)
    acc_cases = []
    for i in range(60):
        t = helpers.rand_seq(rng, int(rng.randint(40, 300)))
        a = int(rng.randint(0, len(t) - 20)); q = helpers.mutate(rng, t[a:a + int(rng.randint(10, 120))], 0.15) or "ACGT"
        r = cport.hw_align(q, t, len(q))
        acc_cases.append({"cigar": r["cigar"], "accuracy": float(ref.cigar_to_accuracy(r["cigar"])).hex()})
    out["cigar_to_accuracy"] = acc_cases

    with open(os.path.join(HERE, "classify_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote classify_golden.json:", sum(len(w["instances"]) for w in out["worlds"]), "instances,",
          len(seg_cases), "segment cases")


if __name__ == "__main__":
    main()
