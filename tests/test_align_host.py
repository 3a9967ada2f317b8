"""CPU: the host steps in front of the chainers against vectors produced by the
reference's own functions (tests/golden/make_align_golden.py): seed file parsing, hit lines -> Mem tuples, instance
splitting with the coverage upper bound, reverse complement."""
import gzip
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align_golden.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def rows(mems):
    return [[m.x, m.y, m.c, m.d, m.val, m.j, m.exon_part_id] for m in mems]


@pytest.mark.parametrize("compress", [True, False])
def test_read_seeds(gold, tmp_path, compress):
    from ultra_amd import seed_wrapper
    path = str(tmp_path / "seeds.txt")
    with (gzip.open(path, "wb") if compress else open(path, "wb")) as fh:
        fh.write(gold["seed_text"].encode())
    got = [list(t) for t in seed_wrapper.read_seeds(path)]
    assert got == gold["parsed"]
    assert any(not g[1] and not g[3] for g in got) and any(not g[1] and g[3] for g in got)


def test_seed_format_round_trip(gold, tmp_path):
    from ultra_amd import align, seed_wrapper
    recs = []
    for acc, hits, acc_rc, hits_rc in gold["parsed"]:
        fw = [m for ms in align.get_mems_from_input(hits).values() for m in ms]
        rc = [m for ms in align.get_mems_from_input(hits_rc).values() for m in ms]
        recs.append((acc, fw, rc))
    path = str(tmp_path / "again.gz")
    seed_wrapper.write_seeds(path, recs)
    for (acc, fw, rc), (acc2, hits, acc_rc, hits_rc) in zip(recs, seed_wrapper.read_seeds(path)):
        assert acc2 == acc and acc_rc == acc + " Reverse"
        assert sorted(m[:5] for ms in align.get_mems_from_input(hits).values() for m in ms) == sorted(m[:5] for m in fw)
        assert sorted(m[:5] for ms in align.get_mems_from_input(hits_rc).values() for m in ms) == sorted(m[:5] for m in rc)


def test_mems_and_bounds(gold):
    from ultra_amd import align
    intron = {int(k): v for k, v in gold["max_intron_chr"].items()}
    n_split = 0
    for run in gold["runs"]:
        g = run["params"]["max_global_intron"]
        for (acc, hits, acc_rc, hits_rc), want in zip(gold["parsed"], run["reads"]):
            mems, mems_rc = align.get_mems_from_input(hits), align.get_mems_from_input(hits_rc)
            assert {str(k): rows(v) for k, v in mems.items()} == want["mems"]
            assert {str(k): rows(v) for k, v in mems_rc.items()} == want["mems_rc"]
            assert list(map(str, mems.keys())) == list(want["mems"].keys())      # dict order feeds the stable sort
            ub = align.annotate_guaranteed_optimal_bound(mems, False, intron, g)
            ub_rc = align.annotate_guaranteed_optimal_bound(mems_rc, True, intron, g)
            got = [[k[0], k[1], v[0], v[1], rows(v[2])] for k, v in list(ub.items()) + list(ub_rc.items())]
            assert got == want["bounds"]
            n_split += sum(1 for b in got if b[1] > 0)
    assert n_split > 10


def test_reverse_complement(gold):
    from ultra_amd import align
    seqs = [r[:60] for r in gold["reads"][:8]] + ["ACGTUNacgtunRYKMSWBVDHXrykmswbvdh"]
    assert [align.reverse_complement(s) for s in seqs] == gold["revcomp"]
    with pytest.raises(KeyError):
        align.reverse_complement("ACG-T")
