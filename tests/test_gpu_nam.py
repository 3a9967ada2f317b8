"""GPU parity of NAM chaining through the C ABI (ultra_chain_nams_batch) and the
reference-named shims: against the golden vectors the reference's own solvers produced
(tests/golden/nam_golden.json), and against the oracle on seeded random instances - DP vectors, optimal value
and every co-optimal chain, bit-exact (integer work)."""
import json
import os
from collections import namedtuple

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nam_golden.json")
Mem = namedtuple("Mem", ["x", "y", "c", "d", "val", "j", "exon_part_id"])


def _instance(rng, n, style):
    """Seed lists the way get_mems_from_input builds them: sorted by reference position."""
    if style == "colinear":
        x = np.sort(rng.integers(0, 40 * n + 50, n))
        c = np.sort(rng.integers(0, 12 * n + 30, n))
        ln = rng.integers(10, 60, n)
    elif style == "ties":
        x = np.sort(rng.integers(0, 8 * n + 10, n))
        c = rng.integers(0, 3 * n + 10, n)
        ln = rng.integers(1, 8, n) * 5
    else:
        x = np.sort(rng.integers(0, 100000, n))
        c = rng.integers(0, 3000, n)
        ln = rng.integers(1, 400, n)
    d = c + ln - 1
    y = x + ln - 1
    return y.astype(np.int32), c.astype(np.int32), d.astype(np.int32)


def _mems(y, c, d):
    return [Mem(int(y[j]) - (int(d[j]) - int(c[j])), int(y[j]), int(c[j]), int(d[j]), int(d[j]) - int(c[j]) + 1, j, "e")
            for j in range(len(c))]


def test_golden_vectors(ctx):
    from ultra_amd import colinear_solver
    cases = json.load(open(GOLD))["cases"]
    insts, introns, solvers, wants = [], [], [], []
    for rec in cases:
        mems = _mems(rec["y"], rec["c"], rec["d"])
        for key, want in rec.items():
            if key.startswith("quad_"):
                insts.append(mems); introns.append(int(key[5:])); solvers.append(1); wants.append(want)
            elif key == "nlogn":
                insts.append(mems); introns.append(0); solvers.append(2); wants.append(want)
    assert len(insts) > 250
    for solver in (1, 2):
        sel = [i for i, s in enumerate(solvers) if s == solver]
        got = colinear_solver.read_coverage_batch([insts[i] for i in sel], [introns[i] for i in sel], solver=solver, ctx=ctx)
        for i, (sols, value) in zip(sel, got):
            assert value == wants[i]["value"], (i, solver)
            assert [[m.j for m in s] for s in sols] == wants[i]["chains"], (i, solver)


@pytest.mark.parametrize("solver", [1, 2])
def test_random_vs_oracle(ctx, solver):
    from oracle import cport
    rng = np.random.default_rng(77 + solver)
    ys, cs, ds, off, introns = [], [], [], [0], []
    sizes = [0, 1, 2, 3, 63, 64, 65, 89, 90, 127, 128, 129, 255, 256, 257, 300, 511, 512, 1023, 1500, 2500]
    for n in sizes:
        for style in ("colinear", "ties", "random"):
            y, c, d = _instance(rng, n, style)
            ys.append(y); cs.append(c); ds.append(d); off.append(off[-1] + n)
            introns.append(int(rng.choice([200, 5000, 10 ** 9])))
    C, tr, value = ctx.chain_nams_batch(off, np.concatenate(ys), np.concatenate(cs), np.concatenate(ds), introns, solver=solver)
    for i in range(len(ys)):
        if solver == 1:
            wc, wt = cport.nam_quadratic(ys[i], cs[i], ds[i], introns[i])
        else:
            wc, wt = cport.nam_nlogn(cs[i], ds[i])
        lo, hi = off[i] + i, off[i + 1] + i + 1
        assert C[lo:hi].tolist() == wc, (i, len(cs[i]))
        assert tr[lo:hi].tolist() == wt, (i, len(cs[i]))
        assert int(value[i]) == max(wc)


def test_solver_switch_and_shims(ctx):
    """solver 0 = align.py:448: fewer than 90 NAMs -> quadratic with the intron limit, otherwise n log n."""
    from oracle import cport
    from ultra_amd import colinear_solver
    rng = np.random.default_rng(5)
    for n in (89, 90):
        y, c, d = _instance(rng, n, "random")
        mems = _mems(y, c, d)
        (sols, value), = colinear_solver.read_coverage_batch([mems], [300], ctx=ctx)
        wc, wt = cport.nam_quadratic(y, c, d, 300) if n < 90 else cport.nam_nlogn(c, d)
        chains, best = cport.all_optimal_chains(wc, wt)
        assert value == best and [[m.j for m in s] for s in sols] == chains
    y, c, d = _instance(rng, 40, "ties")
    mems = _mems(y, c, d)
    sols, value = colinear_solver.read_coverage(mems, 1000)
    chains, best = cport.all_optimal_chains(*cport.nam_quadratic(y, c, d, 1000))
    assert value == best and [[m.j for m in s] for s in sols] == chains and all(isinstance(m, Mem) for s in sols for m in s)
    sols, value = colinear_solver.n_logn_read_coverage(mems)
    chains, best = cport.all_optimal_chains(*cport.nam_nlogn(c, d))
    assert value == best and [[m.j for m in s] for s in sols] == chains
    # no NAMs: the reference returns one empty chain of value 0
    assert colinear_solver.read_coverage([], 10) == ([[]], 0)


def test_limits_and_bad_input(ctx):
    from oracle import cport
    from ultra_amd.lib import UltraError
    rng = np.random.default_rng(9)
    y, c, d = _instance(rng, 8191, "colinear")
    C, tr, value = ctx.chain_nams_batch([0, 8191], None, c, d, None, solver=2)
    wc, wt = cport.nam_nlogn(c, d)
    assert C.tolist() == wc and tr.tolist() == wt
    y, c, d = _instance(rng, 6000, "random")
    C, tr, value = ctx.chain_nams_batch([0, 6000], y, c, d, [20000], solver=1)
    wc, wt = cport.nam_quadratic(y, c, d, 20000)
    assert C.tolist() == wc and tr.tolist() == wt
    with pytest.raises(UltraError):
        ctx.chain_nams_batch([0, 2], [5, 9], [3, 8], [5, 7], [100], solver=1)      # d < c
    with pytest.raises(UltraError):
        ctx.chain_nams_batch([0, 2], None, [3, 8], [5, 9], None, solver=1)         # quadratic needs y
    with pytest.raises(UltraError):
        y, c, d = _instance(rng, 8192, "colinear")
        ctx.chain_nams_batch([0, 8192], None, c, d, None, solver=2)                # beyond the LDS limit
