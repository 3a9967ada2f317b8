"""NAM chaining (row f1) throughput: ultra_chain_nams_batch on the GPU, host buffers in and out, next to the
oracle's C restatement on one host core.  Prints one JSON line.

    python scripts/bench_nam.py [--instances 200000] [--cpu-seconds 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make(rng, n_inst):
    """Instance sizes like seeded long reads: most instances a few dozen NAMs, a tail of repeat-rich ones."""
    sizes = np.minimum(np.maximum(rng.lognormal(3.3, 0.9, n_inst).astype(np.int64), 1), 4000)
    off = np.zeros(n_inst + 1, np.int64); np.cumsum(sizes, out=off[1:])
    tot = int(off[-1])
    inst = np.repeat(np.arange(n_inst), sizes)
    rank = np.arange(tot) - off[inst]
    ln = rng.integers(12, 80, tot)
    c = (rank * 25 + rng.integers(-40, 41, tot)).clip(0)          # roughly colinear with noise
    noise = rng.random(tot) < 0.2
    c[noise] = rng.integers(0, 25 * sizes[inst][noise] + 1)
    x = rank * 400 + rng.integers(0, 100, tot)
    y = x + ln - 1
    # sorted by y inside each instance
    order = np.lexsort((y, inst))
    return off, y[order].astype(np.int32), c[order].astype(np.int32), (c + ln - 1)[order].astype(np.int32), sizes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=200000)
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    off, y, c, d, sizes = make(rng, a.instances)
    introns = np.full(a.instances, 200000, np.int64)
    from oracle import cport
    t0 = time.perf_counter(); done = 0
    while time.perf_counter() - t0 < a.cpu_seconds and done < a.instances:
        lo, hi = off[done], off[done + 1]
        if hi - lo < 90:
            cport.nam_quadratic(y[lo:hi], c[lo:hi], d[lo:hi], 200000)
        else:
            cport.nam_nlogn(c[lo:hi], d[lo:hi])
        done += 1
    cpu_rate = done / (time.perf_counter() - t0)
    from ultra_amd import lib
    ctx = lib.Context(0)
    ctx.chain_nams_batch(off, y, c, d, introns)
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        C, tr, value = ctx.chain_nams_batch(off, y, c, d, introns)
        ts.append(time.perf_counter() - t0)
    # spot check against the oracle
    for i in rng.integers(0, a.instances, 200):
        lo, hi = off[i], off[i + 1]
        wc, wt = cport.nam_quadratic(y[lo:hi], c[lo:hi], d[lo:hi], 200000) if hi - lo < 90 else cport.nam_nlogn(c[lo:hi], d[lo:hi])
        assert C[lo + i:hi + i + 1].tolist() == wc and tr[lo + i:hi + i + 1].tolist() == wt, i
    t = min(ts)
    print(json.dumps({"metric": "nam_chain_instances_per_second", "value": a.instances / t, "ms": t * 1e3,
                      "instances": a.instances, "nams": int(off[-1]), "median_n": int(np.median(sizes)),
                      "frac_nlogn": float((sizes >= 90).mean()), "includes": "H2D + kernels + D2H",
                      "cpu_port_1core": cpu_rate, "cpu_sample": "%d instances" % done, "checked_vs_oracle": 200}))


if __name__ == "__main__":
    main()
