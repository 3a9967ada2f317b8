"""CPU: pins the MAM-stage oracle (oracle/classify_port.py) against golden vectors produced by running
the reference's own classify_read_with_mams.py with edlib.align bound to the same CPU Myers oracle
(tests/golden/make_classify_golden.py).  Everything the reference does around the edlib call -
accuracy, MAM emission, candidate enumeration, partial-segment logic, filters, sort, solver switch
and chaining - is compared exactly (FP64 values bit for bit, mam_id strings included)."""
import json
import os

import pytest

from tests import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "classify_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_main_matches_reference_vectors(gold):
    from oracle import classify_port
    n = 0
    for w in gold["worlds"]:
        idx = helpers.load_golden_world(w["world"])
        for inst in w["instances"]:
            sol = helpers.golden_solution(inst["solution"])
            gaps, value, mams = classify_port.classify(sol, *idx, inst["read"], None, inst["min_acc"])
            assert [int(g) for g in gaps] == inst["non_covered_regions"]
            assert float(value).hex() == inst["value"]
            assert helpers.mam_rows(mams) == inst["mam_solution"]
            n += 1
    assert n >= 140


def test_add_segment_to_mam_matches_reference_vectors(gold):
    from oracle import classify_port
    shapes = 0
    for c in gold["add_segment_to_mam"]:
        out = []
        classify_port.emit_mams(c["read"], 3, c["seq"], c["e_start"], c["e_stop"], c["segm_id"], out, c["min_acc"], c["label"])
        assert helpers.mam_rows(out) == c["mams"]
        shapes += len(out)
    assert shapes > 100


def test_accuracy_closed_form_matches_reference_cigar_parser(gold):
    """(L - ED + I) / (L + I) == the reference's cigar_to_accuracy on the same cigar, bit for bit."""
    import re
    for c in gold["cigar_to_accuracy"]:
        ops = [(int(n), t) for n, t in re.findall(r"(\d+)([=XID])", c["cigar"])]
        ins = sum(n for n, t in ops if t == "I")
        ed = sum(n for n, t in ops if t != "=")
        span = sum(n for n, t in ops if t != "I")
        assert float((span - ed + ins) / float(span + ins)).hex() == c["accuracy"]
