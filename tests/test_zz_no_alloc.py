"""The one entry point that allocates nothing: ultra_chain_mams_batch_noalloc (fixed device scratch inside the
library image) against ultra_chain_mams_batch, bit for bit, through the C-ABI.  Two calls in a row, the second with a
smaller input: the fixed buffer keeps the first call's bytes, and nothing of them may leak into the second result.
Named so that pytest collects it after every other test file."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_chain import _batch

pytestmark = pytest.mark.gpu


def _both(ctx, batch, **kw):
    a = ctx.chain_mams_batch(*batch, **kw)
    b = ctx.chain_mams_batch(*batch, noalloc=True, **kw)
    off = batch[0]
    assert np.array_equal(a[1], b[1])                       # chain lengths
    assert a[2].tobytes() == b[2].tobytes()                 # FP64 value bits
    for i in range(len(off) - 1):
        o = int(off[i])
        assert np.array_equal(a[0][o:o + a[1][i]], b[0][o:o + b[1][i]]), i
    return a


def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row(ctx):
    rng = np.random.RandomState(71)
    # first call: every solver (lane-per-instance, small / big quadratic, n log n), empty instances in between
    sizes = [int(s) for s in rng.choice([0, 1, 2, 5, 17, 47, 48, 49, 64, 130, 199, 200, 201, 256, 300, 700], size=600)]
    big = _batch(rng, sizes, [100, 300, 1000, 3000])
    _both(ctx, big, solver=0)
    # second call: smaller, other contents - stale bytes of the first call lie behind and between its arrays
    small = _batch(rng, [3, 0, 201, 7, 48, 0, 1], [80, 400])
    _both(ctx, small, solver=0)
    _both(ctx, small, solver=2, overlap_large=3)
    _both(ctx, small, solver=1, overlap_small=7)
    # and against the oracle, so that "equal" is not "equally wrong"
    from tests.test_gpu_chain import _compare
    from oracle import cport  # noqa: F401
    idx, ln, val = ctx.chain_mams_batch(*small, solver=0, noalloc=True)
    off, x, y, c, d, v = small
    for i in range(len(off) - 1):
        o, e = int(off[i]), int(off[i + 1])
        if e == o:
            assert ln[i] == 0 and val[i] == -1.0
            continue
        want, wv = (cport.chain_nlogn(x[o:e], y[o:e], c[o:e], d[o:e], v[o:e], 5) if e - o > 200
                    else cport.chain_quadratic(c[o:e], d[o:e], v[o:e], 20))
        assert idx[o:o + ln[i]].tolist() == want and np.float64(val[i]).tobytes() == np.float64(wv).tobytes()


def _free_device_bytes():
    """hipMemGetInfo through the HIP runtime the library has loaded."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_allocates_no_device_memory(ctx):
    """Free device memory is the same before and after calls that run every solver, the n log n one included (whose
    tree arrays are the largest piece of scratch), on inputs larger than anything the context has seen."""
    rng = np.random.RandomState(73)
    warm = _batch(rng, [4, 30], [100])
    ctx.chain_mams_batch(*warm, noalloc=True)               # module load, first use of the streams
    before = _free_device_bytes()
    sizes = [int(s) for s in rng.choice([1, 9, 48, 49, 199, 201, 450, 900, 2500], size=120)]
    batch = _batch(rng, sizes, [300, 3000])
    for kw in (dict(solver=0), dict(solver=2), dict(solver=1)):
        ctx.chain_mams_batch(*batch, noalloc=True, **kw)
    assert _free_device_bytes() == before
    # the allocating entry point on the same input, in a context that has no buffers yet, does take memory (the check can
    # see an allocation)
    from ultra_amd import lib
    fresh = lib.Context(0)
    try:
        at_start = _free_device_bytes()
        fresh.chain_mams_batch(*batch, solver=2)
        assert _free_device_bytes() < at_start
    finally:
        fresh.close()


def test_noalloc_rejects_what_does_not_fit(ctx):
    from ultra_amd import lib
    cap = ctx.lib.ultra_chain_noalloc_capacity()
    assert cap == 8 << 20
    rng = np.random.RandomState(72)
    n = cap // 28 + 1000                     # one instance list that cannot fit 28 B per MAM
    sizes = [40] * (n // 40)
    too_big = _batch(rng, sizes[:50], [300])
    # tile the small batch's arrays instead of generating 300 000 MAMs one by one
    reps = len(sizes) // 50 + 1
    off = np.concatenate([[0], np.cumsum(np.tile(np.diff(too_big[0]), reps))]).astype(np.int64)
    arrs = [np.tile(a, reps) for a in too_big[1:]]
    with pytest.raises(lib.UltraError) as err:
        ctx.chain_mams_batch(off, *arrs, noalloc=True)
    assert err.value.code == lib.ULTRA_E_CAPACITY
    # the allocating entry point takes the same batch, and the fixed buffer still works afterwards
    ctx.chain_mams_batch(off, *arrs)
    _both(ctx, _batch(rng, [5, 60, 0, 210], [200]), solver=0)


def test_noalloc_empty_batch(ctx):
    idx, ln, val = ctx.chain_mams_batch(np.zeros(1, np.int64), [], [], [], [], [], noalloc=True)
    assert len(ln) == 0
