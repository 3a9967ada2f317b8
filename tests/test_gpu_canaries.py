"""GPU: the hot path under debug canaries (ROCm has no compute-sanitizer, so every device buffer of
the library can be placed between guard regions - ULTRA_CANARIES=1 - and an out-of-bounds store of a kernel shows up as a
damaged guard).  The mode is read once per process, so the work runs in a child process: a device-planned and a
host-planned MAM batch with flanks and overflow-prone instances, a raw Myers batch over every size class including the
long one, the MAM and NAM chainers and the affine alignment (the parity tests of those rows, called in the child);
results are still compared with the oracle there, then every guard is read back.  The self-test overruns a scratch buffer by one byte and requires the check to notice."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, struct, sys
import numpy as np
sys.path.insert(0, %(root)r)
from ultra_amd import hwq
hwq.export_hw_queues()
from oracle import classify_port
from tests import helpers
from ultra_amd import classify_read_with_mams as stage
from ultra_amd import colinear_solver, index as native, lib, synth
from ultra_amd import align as align_mod

so = lib.load()
def guards(self_test=0):
    nb, nd = C.c_int64(0), C.c_int64(0)
    rc = so.ultra_debug_canaries(C.byref(nb), C.byref(nd), self_test)
    assert rc == 0, rc
    return nb.value, nd.value

ctx = lib.Context(0)
assert guards(1)[1] == 0                      # the check sees a one-byte overrun of its own scratch buffer
def same(a, b):
    return (a[0] == b[0] and struct.pack("d", float(a[1])) == struct.pack("d", float(b[1]))
            and tuple(tuple(m) for m in a[2]) == tuple(tuple(m) for m in b[2]))
for preset, n, seed in (("ont_cdna", 333, 5), ("ont_drna", 97, 6), ("isoseq", 61, 7)):
    w = synth.make_world(seed=90 + seed, n_genes=70, n_chr=2)
    idx = w.index()
    insts = synth.make_instances(w, n, seed=seed, preset=preset, flank_start_prob=0.35, flank_end_prob=0.35)
    pairs = [(sol, read) for read, sol, _ in insts]
    min_acc = synth.PRESETS[preset]["min_acc"]
    got = stage.main_batch(pairs, *idx, min_acc=min_acc, ctx=ctx)           # host-planned
    ix = native.NativeIndex(*idx); dix = native.DeviceIndex(ix)
    flat = native.flatten_instances(pairs)
    ctx.mam_stage_nams(dix, *flat); ctx.mam_run(min_acc)                     # device-planned
    dev = stage.results_from_names(ctx.mam_fetch(want_names=True), np.diff(flat[5]).tolist())
    for i, (read, sol, _) in enumerate(insts):
        assert same(got[i], dev[i])
        if i %% 5 == 0:
            assert same(classify_port.classify(sol, *idx, read, None, min_acc), got[i])
    assert guards()[1] == 0, preset
    dix.close(); ix.close()
# raw Myers batches over every size class (lane, lane-group, whole-wave, long), odd lengths, many-location targets
from tests.test_gpu_myers import _check
rng = np.random.RandomState(3)
_check(ctx, helpers.make_pairs(rng, 1500, (1, 300), (20, 900), kmode="mixed"))
_check(ctx, helpers.make_pairs(rng, 90, (257, 2100), (300, 2600), kmode="ref"))
_check(ctx, helpers.make_pairs(rng, 10, (2100, 9000), (1500, 9000), kmode="ref", rates=(0.02, 0.1)))
_check(ctx, helpers.make_pairs(rng, 2, (33000, 34000), (35000, 40000), kmode="ref", rates=(0.02,), planted=1.0))
_check(ctx, helpers.make_pairs(rng, 300, (5, 30), (200, 1000), kmode="ref", alpha="A"))
_check(ctx, helpers.make_pairs(rng, 60, (9, 200), (100, 500), kmode="ref"), want_ops=True)
assert guards()[1] == 0, "myers"
# the chainers (lane-per-instance, quadratic, n log n; MAM and NAM objective) and the affine alignment: the parity tests
# of those rows, run here under the guards
from tests import test_gpu_chain, test_gpu_nam, test_gpu_affine
test_gpu_chain.test_reference_switch_rule(ctx)
test_gpu_chain.test_lane_per_instance_kernel(ctx)
test_gpu_chain.test_forced_quadratic_beyond_200(ctx)
test_gpu_chain.test_tree_large_and_thresholds(ctx)
assert guards()[1] == 0, "MAM chaining"
test_gpu_nam.test_random_vs_oracle(ctx, 1)
test_gpu_nam.test_random_vs_oracle(ctx, 2)
assert guards()[1] == 0, "NAM chaining"
test_gpu_affine.test_small_and_ragged_pairs(ctx)
test_gpu_affine.test_read_sized_pairs(ctx)
assert guards()[1] == 0, "affine alignment"
nb, nd = guards()
print("canaries ok: %%d device buffers guarded, %%d damaged" %% (nb, nd))
ctx.close()
'''


def test_hot_path_leaves_every_guard_intact():
    env = dict(os.environ)
    env["ULTRA_CANARIES"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "canaries ok" in p.stdout


def test_canary_check_is_off_by_default(ctx):
    import ctypes as C
    from ultra_amd import lib
    if os.environ.get("ULTRA_CANARIES", "0") not in ("", "0"):
        pytest.skip("the session itself runs under canaries")
    nb, nd = C.c_int64(0), C.c_int64(0)
    assert lib.load().ultra_debug_canaries(C.byref(nb), C.byref(nd), 0) == -6      # ULTRA_E_STATE
