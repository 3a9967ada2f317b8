"""CPU: the native candidate enumeration (csrc/plan.hip through the C ABI, no GPU call) builds the same batch as the
Python port `classify_read_with_mams.prepare_batch`, which is pinned by the reference's golden vectors
(tests/test_oracle_classify.py).  Compared per instance: candidate order, kinds, coordinates, sequences, the edlib
call shapes (query, target, k) and the sequence identity classes."""
import json
import os

import numpy as np
import pytest

from tests import helpers
from ultra_amd import classify_read_with_mams as stage
from ultra_amd import index as native
from ultra_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pairs_of(packed, p):
    if p < 0:
        return None
    a = packed["arena"]
    q = bytes(a[packed["q_off"][p]:packed["q_off"][p] + packed["q_len"][p]])
    t = bytes(a[packed["t_off"][p]:packed["t_off"][p] + packed["t_len"][p]])
    return q, t, int(packed["k"][p])


def _canon(packed, plans=None):
    """Per instance: list of (kind, e_start, e_stop, seq_len, pair, contained pair, seq class) with pairs by content."""
    out = []
    off = packed["inst_cand_off"]
    for i in range(len(off) - 1):
        rows = []
        for c in range(int(off[i]), int(off[i + 1])):
            rows.append((int(packed["cand_kind"][c]), int(packed["cand_e_start"][c]), int(packed["cand_e_stop"][c]),
                         int(packed["cand_seq_len"][c]), _pairs_of(packed, int(packed["cand_pair"][c])),
                         _pairs_of(packed, int(packed["cand_pair_contained"][c])), int(packed["cand_seq_id"][c])))
        out.append((rows, int(packed["inst_read_len"][i]), int(packed["inst_first_part_stop"][i]),
                    int(packed["inst_last_part_start"][i])))
    return out


def _compare(idx, instances):
    ref = stage.prepare_batch(instances, *idx)
    ix = native.NativeIndex(*idx)
    try:
        plan = ix.plan(instances)
        assert plan.n_inst == ref.n_inst
        assert _canon(plan.packed) == _canon(ref.packed)
        assert plan.cells >= ref.cells  # sequences are shared by location, not by content: never fewer pairs
        # names of the MAMs a candidate would produce
        c = 0
        for p in ref.plans:
            for cand in p.cands:
                assert plan.mam_id(c, stage.LABELS[cand.kind]) == str(cand.ident) + stage.LABELS[cand.kind]
                assert int(plan.cand_chr[c]) == cand.chr_id
                c += 1
        plan.close()
    finally:
        ix.close()


@pytest.mark.parametrize("preset,kw", [("ont_cdna", {}), ("isoseq", {}),
                                        ("ont_drna", {"flank_start_prob": 0.5, "flank_end_prob": 0.5})])
def test_native_plan_equals_python_plan(preset, kw):
    w = synth.make_world(seed=31, n_genes=60)
    insts = synth.make_instances(w, 150, seed=8, preset=preset, **kw)
    _compare(w.index(), [(sol, read) for read, sol, _ in insts])


def test_native_plan_small_exons_and_split_chains():
    w = synth.make_world(seed=32, n_genes=40, small_exon_frac=0.4, split_frac=0.7)
    insts = synth.make_instances(w, 120, seed=9, preset="ont_cdna", flank_start_prob=0.4, flank_end_prob=0.4)
    _compare(w.index(), [(sol, read) for read, sol, _ in insts])


def test_native_plan_on_reference_golden_worlds():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "classify_golden.json")))
    for world in gold["worlds"]:
        idx = helpers.load_golden_world(world["world"])
        instances = [(helpers.golden_solution(i["solution"]), i["read"]) for i in world["instances"]]
        _compare(idx, instances)


def test_python_slice_semantics_of_partial_pieces():
    """A NAM that starts before its segment or ends after it makes the reference slice with a negative start / a stop
    beyond the end (classify_read_with_mams.py:486, :494): seq[-3:] and seq[:len+2] must come out as Python's."""
    from array import array
    key = lambda c, a, b: array('L', [c, a, b]).tobytes()
    seq = "ACGTACGTACGTACGTACGTACGTACGTAC"  # 30 bases, segment [100, 130)
    seg, flank = {key(1, 100, 130): seq}, {}
    p2s = {key(1, 90, 140): array('L', [1, 100, 130])}
    s2g = {key(1, 100, 130): {"g"}}
    g2s = {"g": array('L', [1, 100, 130])}
    idx = (seg, flank, p2s, s2g, g2s)
    read = "ACGTACGTACGTACGTACGTACGTACGTACGGGG"
    for x, y in [(97, 120), (110, 133), (95, 136), (100, 129), (120, 125)]:
        sol = [synth.Mem(x, y, 0, y - x, y - x + 1, 0, "1^90^140")]
        _compare(idx, [(sol, read)])


def test_missing_index_entries_fail_loudly():
    from array import array
    from ultra_amd import lib
    key = lambda c, a, b: array('L', [c, a, b]).tobytes()
    idx = ({}, {}, {key(1, 90, 140): array('L', [1, 100, 130])}, {key(1, 100, 130): {"g"}}, {"g": array('L', [1, 100, 130])})
    ix = native.NativeIndex(*idx)
    with pytest.raises(lib.UltraError):
        ix.plan([([synth.Mem(100, 120, 0, 20, 21, 0, "1^90^140")], "ACGTACGTACGTACGTACGTA")])  # no sequence for the segment
    ix.close()
    with pytest.raises(KeyError):
        native.NativeIndex({}, {}, {}, {key(1, 100, 130): {"g"}}, {})
