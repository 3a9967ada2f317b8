"""GPU parity: chaining kernels (through the C ABI) vs the CPU oracle: chain indices and FP64 value bits."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


def _batch(rng, sizes, readlens):
    xs, ys, cs, ds, vs, off = [], [], [], [], [], [0]
    for n in sizes:
        x, y, c, d, v = helpers.make_mams(rng, n, int(rng.choice(readlens)), rng.rand() < 0.5)
        xs.append(x); ys.append(y); cs.append(c); ds.append(d); vs.append(v)
        off.append(off[-1] + n)
    cat = lambda l, dt: np.concatenate(l).astype(dt) if l else np.zeros(0, dt)
    return np.array(off, np.int64), cat(xs, np.int32), cat(ys, np.int32), cat(cs, np.int32), cat(ds, np.int32), cat(vs, np.float64)


def _compare(ctx, off, x, y, c, d, v, solver, **kw):
    from oracle import cport
    idx, ln, val = ctx.chain_mams_batch(off, x, y, c, d, v, solver=solver, **kw)
    for i in range(len(off) - 1):
        o, e = int(off[i]), int(off[i + 1])
        n = e - o
        use_tree = solver == 2 or (solver == 0 and n > 200)
        if n == 0:
            assert ln[i] == 0 and val[i] == -1.0
            continue
        if use_tree:
            want, wv = cport.chain_nlogn(x[o:e], y[o:e], c[o:e], d[o:e], v[o:e], kw.get("overlap_large", 5))
        else:
            want, wv = cport.chain_quadratic(c[o:e], d[o:e], v[o:e], kw.get("overlap_small", 20))
        assert idx[o:o + ln[i]].tolist() == want, (i, n)
        assert np.float64(val[i]).tobytes() == np.float64(wv).tobytes(), (i, n, val[i], wv)


def test_reference_switch_rule(ctx):
    rng = np.random.RandomState(1)
    sizes = [int(s) for s in rng.choice([0, 1, 2, 3, 5, 17, 63, 64, 65, 128, 199, 200, 201, 256, 257, 400, 1000], size=400)]
    _compare(ctx, *_batch(rng, sizes, [100, 300, 1000, 3000]), solver=0)


def test_lane_per_instance_kernel(ctx):
    """Instances of at most 48 MAMs take the lane-per-instance kernel, larger ones in the same batch the
    wavefront-per-instance one: sizes around that limit, mixed within a wavefront, short reads (many ties and
    overlaps) and two overlap thresholds."""
    rng = np.random.RandomState(6)
    sizes = [int(s) for s in rng.randint(0, 60, size=1500)] + [47, 48, 49, 48, 0, 1, 48, 49]
    _compare(ctx, *_batch(rng, sizes, [40, 100, 300]), solver=0)
    _compare(ctx, *_batch(rng, sizes[:400], [60, 200]), solver=1, overlap_small=3)


def test_forced_quadratic_beyond_200(ctx):
    rng = np.random.RandomState(2)
    _compare(ctx, *_batch(rng, [255, 256, 257, 300, 700, 1500], [1000, 5000]), solver=1)


def test_forced_tree_small_instances(ctx):
    rng = np.random.RandomState(3)
    sizes = [int(s) for s in rng.randint(1, 130, size=300)]
    _compare(ctx, *_batch(rng, sizes, [100, 200, 1000]), solver=2)


def test_tree_large_and_thresholds(ctx):
    rng = np.random.RandomState(4)
    _compare(ctx, *_batch(rng, [2047, 2048, 4000, 8191], [2000, 15000]), solver=2)
    _compare(ctx, *_batch(rng, [50, 300, 900], [500]), solver=2, overlap_large=1)
    _compare(ctx, *_batch(rng, [50, 150, 200], [500]), solver=1, overlap_small=5)


def test_limits_fail_loudly(ctx):
    from ultra_amd import lib
    rng = np.random.RandomState(5)
    b = _batch(rng, [9000], [20000])
    with pytest.raises(lib.UltraError):
        ctx.chain_mams_batch(*b, solver=2)
