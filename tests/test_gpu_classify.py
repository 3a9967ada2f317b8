"""GPU parity of the whole MAM stage: ultra_amd.classify_read_with_mams.main_batch (HIP Myers +
HIP chaining through the C ABI) vs the CPU restatement of the reference's main()."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (a[0] == b[0] and struct.pack("d", float(a[1])) == struct.pack("d", float(b[1]))
            and tuple(tuple(m) for m in a[2]) == tuple(tuple(m) for m in b[2]))


@pytest.mark.parametrize("preset,n", [("ont_cdna", 300), ("isoseq", 120), ("ont_drna", 200)])
def test_main_batch_matches_oracle(ctx, preset, n):
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import synth
    w = synth.make_world(seed=21, n_genes=80)
    insts = synth.make_instances(w, n, seed=4, preset=preset)
    idx = w.index()
    min_acc = synth.PRESETS[preset]["min_acc"]
    got = stage.main_batch([(sol, read) for read, sol, _ in insts], *idx, min_acc=min_acc, ctx=ctx)
    bad = 0
    for (read, sol, _), g in zip(insts, got):
        want = classify_port.classify(sol, *idx, read, None, min_acc)
        if not _same(want, g):
            bad += 1
    assert bad == 0


def test_single_instance_signatures(ctx):
    from oracle import classify_port
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import colinear_solver, edlib_shim, synth
    from oracle import cport
    w = synth.make_world(seed=22, n_genes=20)
    read, sol, _ = synth.make_instances(w, 1, seed=5)[0]
    idx = w.index()
    got = stage.main(sol, *idx, read, None, 0.6)
    assert _same(classify_port.classify(sol, *idx, read, None, 0.6), got)
    # edlib-shaped call
    r = edlib_shim.align("ACGTTGCA", "TTACGTAGCATT", mode="HW", task="path", k=3)
    o = cport.hw_align("ACGTTGCA", "TTACGTAGCATT", 3)
    assert r["editDistance"] == o["editDistance"] and r["locations"] == o["locations"] and r["cigar"] == o["cigar"]
    assert edlib_shim.align("AAAAAAAA", "CCCCCCCCCC", mode="HW", task="path", k=2)["editDistance"] == -1
    # chainers with the reference's signatures
    mams = list(got[2])
    if mams:
        sol_q, val_q, _ = colinear_solver.read_coverage_mam_score(mams, overlap_threshold=20)
        assert len(sol_q) >= 1
