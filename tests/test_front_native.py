"""CPU: phase 1 of the native front end (csrc/front.hip: hit lines -> NAMs per chromosome -> instances with bounds in
the reference's processing order, poly-A trimming) against vectors the reference's own functions produced
(tests/golden/align_golden.json: get_mems_from_input, annotate_guaranteed_optimal_bound) and against the Python shims
those vectors pin (ultra_amd.align).  The GPU phases are compared in tests/test_gpu_front.py."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_golden.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def _records(gold):
    return [(acc, read, None, hits, hits_rc) for (acc, hits, _, hits_rc), read in zip(gold["parsed"], gold["reads"])]


def test_instances_and_bounds_equal_the_reference(gold):
    from ultra_amd import front
    intron = {int(k): v for k, v in gold["max_intron_chr"].items()}
    pb = front.PackedBatch.from_records(_records(gold))
    fr = front.Front(None, max_intron_chr=intron, n_threads=3)
    for run in gold["runs"]:
        g = run["params"]["max_global_intron"]
        insts, _ = fr.prepare(pb, g)
        at = 0
        n_multi = 0
        for r, want in enumerate(run["reads"]):
            # the reference's loop order: sorted(list(ub.items()) + list(ub_rc.items()), key=bound, reverse=True), stable
            order = sorted(want["bounds"], key=lambda b: b[2], reverse=True)
            n_multi += len(order) > 2
            for chr_id, _, bound, is_rc, rows in order:
                read, c, rc, b, limit, nams = insts[at]
                at += 1
                assert (read, c, rc, b) == (r, chr_id, is_rc, bound)
                assert limit == min(intron[chr_id] + 20000, g)
                assert nams == [(row[1], row[2], row[3]) for row in rows]
        assert at == len(insts) and n_multi > 5
    fr.close()


def test_trimmed_reads_and_thread_count_independence(gold):
    from ultra_amd import align, front
    rng = np.random.RandomState(3)
    recs = []
    for i, (acc, read, _, hits, hits_rc) in enumerate(_records(gold)):
        tail = ["", "A" * 25, "T" * 9, "AAAAAAAAAACAAAAAAAAAAAAAAAAAAAAAAA", "G" * 30, "A" * 8 + "T" * 40][i % 6]
        recs.append((acc, read[:int(rng.randint(0, len(read)))] + tail, None, hits, hits_rc))
    recs.append(("one", "A", None, [], []))
    recs.append(("none", "", None, [], []))
    recs.append(("short", "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", None, [], []))
    intron = {int(k): v for k, v in gold["max_intron_chr"].items()}
    pb = front.PackedBatch.from_records(recs)
    outs = []
    for threads in (1, 2, 7):
        fr = front.Front(None, max_intron_chr=intron, n_threads=threads)
        for thr in (20, 5):
            insts, reads = fr.prepare(pb, 1200000, polyA_threshold=thr)
            assert reads == [align.remove_read_polyA_ends(r[1], None, thr, 1)[0] for r in recs]
            outs.append((thr, insts))
        fr.close()
    assert outs[0] == outs[2] == outs[4] and outs[1] == outs[3] == outs[5]


def test_malformed_input_is_an_error_naming_the_read(gold):
    from ultra_amd import front, lib
    intron = {0: 1000, 1: 1000}
    fr = front.Front(None, max_intron_chr=intron)
    ok = "0^100^200\t1\t1\t20\n"
    for bad, what in (("0^100^200\t1\t1\n", "fewer than four"), ("0^100\t1\t1\t20\n", "chr^start^stop"), ("0^100^200^3\t1\t1\t20\n", "chr^start^stop"),
                      ("0^100^200\tx\t1\t20\n", "not an integer"), ("7^100^200\t1\t1\t20\n", "max_intron_chr"), ("\n", "fewer than four")):
        pb = front.PackedBatch.from_records([("a", "ACGT", None, [ok], []), ("b", "ACGT", None, [ok, bad], [])])
        with pytest.raises(lib.UltraError) as e:
            fr.prepare(pb, 1200000)
        assert "read 1" in str(e.value) and what in str(e.value)
    # whitespace as str.split() sees it, a missing final newline, an empty record
    pb = front.PackedBatch.from_records([("a", "ACGT", None, ["  0^100^200  5 \t 3   20   extra\r\n", "1^7^9 1 1 4"], []), ("b", "AC", None, [], [])])
    insts, _ = fr.prepare(pb, 1200000)
    assert [(i[0], i[1], i[2], i[3], i[5]) for i in insts] == [(0, 0, False, 20, [(123, 2, 21)]), (0, 1, False, 4, [(10, 0, 3)])]
    fr.close()
