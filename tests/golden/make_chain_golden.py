"""Generates tests/golden/chain_golden.json and rmq_golden.json by RUNNING THE REFERENCE'S OWN
pure-Python chainers in the build container (the reference cannot travel to the GPU box, the
vectors can).

    python tests/golden/make_chain_golden.py          (needs /root/reference; PYTHONHASHSEED irrelevant)

Reference functions executed (imported from /root/reference, nothing copied):
    modules/colinear_solver.py:276  read_coverage_mam_score(mams, overlap_threshold)
    modules/colinear_solver.py:394  n_logn_read_coverage_mams(mams, overlap_threshold)
    modules/colinear_solver.py:64   make_leafs_power_of_2
    modules/range_query_max_search_tree.py:19,35,137  construct_tree / range_query / update

A fixture is data only: inputs (x, y, c, d, val) and the reference's outputs (chain indices and the
optimal value as an IEEE-754 hex string so that parity is checked bit for bit).
"""
import contextlib
import io
import json
import os
import sys
from collections import namedtuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from modules import colinear_solver as ref_cs                      # noqa: E402
from modules import range_query_max_search_tree as ref_rmq          # noqa: E402
from tests import helpers                                           # noqa: E402

Mam = namedtuple('Mam', ['x', 'y', 'c', 'd', 'val', 'j', "min_segment_length", "mam_id", "ref_chr_id"])


def as_mams(x, y, c, d, val):
    return [Mam(int(x[i]), int(y[i]), int(c[i]), int(d[i]), float(val[i]), i, int(d[i] - c[i] + 1), "m%d" % i, 0)
            for i in range(len(c))]


def run_ref(fn, mams, thr):
    with contextlib.redirect_stdout(io.StringIO()):
        sol, value, _ = fn(mams, overlap_threshold=thr)
    return [m.j for m in sol], float(value).hex()


def chain_cases():
    cases = []
    rng = np.random.RandomState(20241008)
    sizes = list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 199, 200, 201, 256, 257, 511, 513, 1000]
    for n in sizes:
        for ties in (False, True):
            for readlen in ((120, 1000) if n < 100 else (1000, 4000)):
                x, y, c, d, v = helpers.make_mams(rng, n, readlen, ties)
                mams = as_mams(x, y, c, d, v)
                rec = {"n": n, "ties": ties, "x": x.tolist(), "y": y.tolist(), "c": c.tolist(), "d": d.tolist(),
                       "val": [float(t).hex() for t in v]}
                # both solvers on every instance, reference thresholds + one alternative each
                for name, fn, thrs in (("quad", ref_cs.read_coverage_mam_score, (20, 5)),
                                       ("nlogn", ref_cs.n_logn_read_coverage_mams, (5, 1))):
                    if name == "quad" and n > 520:
                        continue  # pure-Python O(n^2): keep generation to seconds
                    for thr in thrs:
                        idx, val = run_ref(fn, mams, thr)
                        rec["%s_thr%d" % (name, thr)] = {"chain": idx, "value": val}
                cases.append(rec)
    # hand-made adversarial instances: identical MAMs, fully nested, exact overlaps at the threshold
    hand = [
        [(0, 10, 0, 9, 10.0)] * 6,
        [(0, 10, 0, 9, 10.0), (20, 30, 10, 19, 10.0), (40, 50, 20, 29, 10.0), (60, 70, 30, 39, 10.0)],
        [(0, 30, 0, 29, 27.0), (5, 35, 10, 39, 30.0), (10, 40, 25, 54, 30.0), (12, 42, 29, 58, 30.0),
         (12, 43, 49, 78, 29.5), (43, 60, 54, 70, 17.0)],
        [(i, i + 8, 2 * i, 2 * i + 7, 8.0) for i in range(40)],
        [(100, 120, 5, 24, 20.0), (100, 120, 5, 24, 18.0), (100, 121, 5, 24, 20.0), (121, 150, 20, 48, 29.0),
         (121, 150, 24, 52, 29.0), (121, 150, 25, 53, 29.0), (121, 150, 44, 72, 29.0)],
    ]
    for rows in hand:
        rows = sorted(rows, key=lambda r: r[1])
        x, y, c, d, v = (np.array([r[i] for r in rows]) for i in range(5))
        mams = as_mams(x, y, c, d, v)
        rec = {"n": len(rows), "ties": True, "x": x.tolist(), "y": y.tolist(), "c": c.tolist(), "d": d.tolist(),
               "val": [float(t).hex() for t in v]}
        for name, fn, thrs in (("quad", ref_cs.read_coverage_mam_score, (20, 5)),
                               ("nlogn", ref_cs.n_logn_read_coverage_mams, (5, 1))):
            for thr in thrs:
                idx, val = run_ref(fn, mams, thr)
                rec["%s_thr%d" % (name, thr)] = {"chain": idx, "value": val}
        cases.append(rec)
    return cases


def rmq_cases():
    """(d[], updates, queries) -> (C_max, j_max) straight from the reference's tree (G2)."""
    out = []
    rng = np.random.RandomState(7)
    for n in [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 255, 257]:
        for rep in range(3):
            dmax = int(rng.choice([5, 30, 300, 3000]))
            d = rng.randint(0, dmax + 1, size=n)
            mams = as_mams(np.zeros(n, int), np.zeros(n, int), d, d, np.ones(n))
            leafs = ref_cs.make_leafs_power_of_2(mams)
            nleaf = len(leafs)
            tree = [0 for _ in range(2 * nleaf)]
            ref_rmq.construct_tree(tree, leafs, nleaf)
            leaf_of = {l.j: i for i, l in enumerate(leafs)}
            n_upd = int(rng.randint(0, n + 1))
            upd_j = [-1] + [int(v) for v in rng.permutation(n)[:n_upd]]
            upd_v = [0.0] + [float(v) for v in np.round(rng.rand(n_upd) * 50, 1)]
            # the solver always sets the start leaf to 0 first (colinear_solver.py:421-423)
            for j, v in zip(upd_j, upd_v):
                ref_rmq.update(tree, leaf_of[j], v, nleaf)
            ql, qr, oc, oj, ok = [], [], [], [], []
            for _ in range(40):
                l = int(rng.randint(-1, dmax + 2))
                r = l - 1 + int(rng.choice([1, 5, 20, dmax + 2]))
                if rng.rand() < 0.3:
                    l = -1
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        cm, jm, _ = ref_rmq.range_query(tree, l, r, nleaf)
                    good = 1
                except (NameError, SystemExit):   # the reference's "BUG" path (sys is never imported there)
                    cm, jm, good = 0.0, 0, 0
                ql.append(l); qr.append(r); oc.append(float(cm).hex()); oj.append(int(jm)); ok.append(good)
            out.append({"d": d.tolist(), "upd_j": upd_j, "upd_v": [float(v).hex() for v in upd_v],
                        "ql": ql, "qr": qr, "c": oc, "j": oj, "ok": ok})
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "chain_golden.json"), "w") as f:
        json.dump({"source": "reference modules/colinear_solver.py run in the build container", "cases": chain_cases()}, f)
    with open(os.path.join(HERE, "rmq_golden.json"), "w") as f:
        json.dump({"source": "reference modules/range_query_max_search_tree.py run in the build container",
                   "cases": rmq_cases()}, f)
    print("wrote", os.path.join(HERE, "chain_golden.json"), os.path.join(HERE, "rmq_golden.json"))
