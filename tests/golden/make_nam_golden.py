"""Generates tests/golden/nam_golden.json by RUNNING THE REFERENCE'S OWN NAM chainers in the build
container:

    modules/colinear_solver.py:169  read_coverage(mems, max_intron)      (align.py:449, < 90 NAMs)
    modules/colinear_solver.py:85   n_logn_read_coverage(mems)           (align.py:452)

    python tests/golden/make_nam_golden.py           (needs /root/reference)

A fixture is data only: NAM coordinates (x, y, c, d) and the reference's answers - the optimal value
and EVERY co-optimal chain as lists of NAM indices, in the reference's order.
"""
import contextlib
import io
import json
import os
import sys
from collections import namedtuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")

from modules import colinear_solver as ref_cs  # noqa: E402

Mem = namedtuple('Mem', ['x', 'y', 'c', 'd', 'val', 'j', "exon_part_id"])


def make_nams(rng, n, readlen, genome, style):
    """NAM instance sorted by y (j = rank), the way align.py:357-388 builds it."""
    rows = []
    for _ in range(n):
        if style == "colinear":       # mostly consistent seeds of one transcript + noise
            L = int(rng.randint(10, 60))
            c = int(rng.randint(0, max(1, readlen - L)))
            x = int(c * (genome / float(readlen)) + rng.randint(-30, 31)) if rng.rand() < 0.8 else int(rng.randint(0, genome))
            x = max(0, x)
        elif style == "ties":         # equal lengths / equal coordinates: many co-optimal chains
            L = int(rng.choice([10, 20]))
            c = int(rng.choice(np.arange(0, max(1, readlen - L), 10)))
            x = int(rng.choice(np.arange(0, genome, 100)))
        else:
            L = int(rng.randint(5, 80))
            c = int(rng.randint(0, max(1, readlen - L)))
            x = int(rng.randint(0, genome))
        rows.append((x, x + L - 1, c, c + L - 1, L))
    rows.sort(key=lambda r: r[1])
    return [Mem(x, y, c, d, v, j, "1^0^%d" % genome) for j, (x, y, c, d, v) in enumerate(rows)]


def run(fn, *args):
    with contextlib.redirect_stdout(io.StringIO()):
        sols, value = fn(*args)
    return [[m.j for m in sol] for sol in sols], int(value)


def main():
    rng = np.random.RandomState(20241008)
    cases = []
    sizes = list(range(1, 25)) + [40, 63, 64, 65, 89, 90, 91, 128, 129, 200, 257, 600]
    for n in sizes:
        for style in ("colinear", "ties", "random"):
            readlen = int(rng.choice([150, 1000, 4000]))
            genome = int(rng.choice([5000, 200000]))
            mems = make_nams(rng, n, readlen, genome, style)
            rec = {"n": n, "style": style, "x": [m.x for m in mems], "y": [m.y for m in mems],
                   "c": [m.c for m in mems], "d": [m.d for m in mems]}
            for max_intron in (genome * 10, 20000 + 3000):
                if n <= 300:
                    sols, value = run(ref_cs.read_coverage, mems, max_intron)
                    rec["quad_%d" % max_intron] = {"chains": sols, "value": value}
            sols, value = run(ref_cs.n_logn_read_coverage, mems)
            rec["nlogn"] = {"chains": sols, "value": value}
            cases.append(rec)
    with open(os.path.join(HERE, "nam_golden.json"), "w") as f:
        json.dump({"source": "reference modules/colinear_solver.py (NAM solvers) run in the build container", "cases": cases}, f)
    print("wrote nam_golden.json:", len(cases), "instances")


if __name__ == "__main__":
    main()
