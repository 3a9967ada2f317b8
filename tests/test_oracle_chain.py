"""CPU: pins the chaining oracle (oracle/chain_port.c) against golden vectors produced by running the
reference's own pure-Python solvers (tests/golden/make_chain_golden.py): chain indices and the FP64
value bit for bit, both solvers, several overlap thresholds, and the range-max search tree alone."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)["cases"]


def _arrays(rec):
    val = np.array([float.fromhex(v) for v in rec["val"]], np.float64)
    return (np.array(rec["x"], np.int32), np.array(rec["y"], np.int32), np.array(rec["c"], np.int32),
            np.array(rec["d"], np.int32), val)


def test_quadratic_solver_matches_reference_vectors():
    from oracle import cport
    n_checked = 0
    for rec in _load("chain_golden.json"):
        x, y, c, d, val = _arrays(rec)
        for key, want in rec.items():
            if not key.startswith("quad_thr"):
                continue
            thr = int(key[len("quad_thr"):])
            chain, value = cport.chain_quadratic(c, d, val, thr)
            assert chain == want["chain"], (rec["n"], thr)
            assert float(value).hex() == want["value"], (rec["n"], thr)
            n_checked += 1
    assert n_checked > 300


def test_nlogn_solver_matches_reference_vectors():
    from oracle import cport
    n_checked = 0
    for rec in _load("chain_golden.json"):
        x, y, c, d, val = _arrays(rec)
        for key, want in rec.items():
            if not key.startswith("nlogn_thr"):
                continue
            thr = int(key[len("nlogn_thr"):])
            chain, value = cport.chain_nlogn(x, y, c, d, val, thr)
            assert chain == want["chain"], (rec["n"], thr)
            assert float(value).hex() == want["value"], (rec["n"], thr)
            n_checked += 1
    assert n_checked > 300


def test_range_max_tree_matches_reference_vectors():
    from oracle import cport
    n_q = 0
    for rec in _load("rmq_golden.json"):
        upd_v = [float.fromhex(v) for v in rec["upd_v"]]
        oc, oj, ok = cport.rmq_run(rec["d"], rec["upd_j"], upd_v, rec["ql"], rec["qr"])
        for i in range(len(rec["ql"])):
            assert int(ok[i]) == rec["ok"][i], (rec["d"], rec["ql"][i], rec["qr"][i])
            if rec["ok"][i]:
                assert float(oc[i]).hex() == rec["c"][i] and int(oj[i]) == rec["j"][i], (rec["d"], rec["ql"][i], rec["qr"][i])
            n_q += 1
    assert n_q > 1500


def test_batch_driver_applies_the_switch_rule():
    from oracle import cport
    from tests import helpers
    rng = np.random.RandomState(3)
    sizes = [0, 1, 5, 200, 201, 300]
    parts = [helpers.make_mams(rng, n, 1000, False) for n in sizes]
    off = np.zeros(len(sizes) + 1, np.int64); np.cumsum(sizes, out=off[1:])
    cat = [np.concatenate([p[i] for p in parts]) for i in range(5)]
    idx, ln, value = cport.chain_batch(off, *cat)
    for i, n in enumerate(sizes):
        o = int(off[i])
        p = parts[i]
        if n == 0:
            assert ln[i] == 0 and value[i] == -1.0
            continue
        want = cport.chain_nlogn(p[0], p[1], p[2], p[3], p[4], 5) if n > 200 else cport.chain_quadratic(p[2], p[3], p[4], 20)
        assert idx[o:o + ln[i]].tolist() == want[0] and float(value[i]).hex() == float(want[1]).hex()
