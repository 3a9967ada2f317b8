"""How much of the MAM stage's output rests on edlib's UNPINNED traceback choice (TEST INFRASTRUCTURE ONLY).

edlib.align(..., task="path") reports the cigar of ONE optimal path of the first location; the reference turns it into
accuracy = (L - ED + I) / (L + I) (classify_read_with_mams.py:71-94), I = inserted query bases on that path, and uses
it for the `accuracy > min_acc` gate and the MAM score max((L - ED) / L, accuracy) * L (:318-349, :385-402).  Which of
the co-optimal paths edlib walks (its tie rule; a Hirschberg split once the traceback state 20 * ceil(m / 64) * L
reaches 1 MiB) cannot be checked in this image - edlib is absent - so this script bounds what it could change: for
every hit of a sample of bench instances, oracle_ins_range gives I_min / I_max over ALL optimal paths; the instance is
then solved three times - with the oracle's own rule (up, left, diagonal: what libultra_hip computes), with I_min
everywhere and with I_max everywhere - and the chosen chains are compared.

    python -m oracle.parity_exposure --reads 6000 --out profiles/r03_parity_exposure.json

Nothing under ultra_amd/'s product path imports this module; it reads no file of the reference.
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import classify_port, cport  # noqa: E402

HIRSCHBERG_BYTES = 1 << 20      # SURVEY.md section 8c, assumption beta

_G = {}


def _accuracy(span, ed, ins):
    return (span - ed + ins) / float(span + ins)


def _solve(inst, idx, min_acc, stats):
    """One instance under the three rules; per-hit figures go to `stats`."""
    read, sol, _ = inst
    cache = {}
    mode = ["alpha"]

    def approx_match(query, target, k):
        key = (query, target, int(k))
        rec = cache.get(key)
        if rec is None:
            res = cport.hw_align(query, target, int(k))
            if res["editDistance"] == -1:
                rec = None, -1, None
            else:
                s, e = res["locations"][0]
                span = e - s + 1
                lo, hi = cport.ins_range(query, target[s:e + 1], res["editDistance"])
                ia = res["insertions"]
                assert lo <= ia <= hi
                rec = res["locations"], res["editDistance"], (span, ia, lo, hi)
                if int(k) > 0:        # k = 0 calls are exact occurrences: one path
                    ed = res["editDistance"]
                    big = 20 * int(math.ceil(len(query) / 64.0)) * span >= HIRSCHBERG_BYTES
                    g = stats["hirschberg" if big else "stored"]
                    g["hits"] += 1
                    g["range"] += hi > lo
                    g["alpha_not_min"] += ia > lo
                    g["alpha_not_max"] += ia < hi
                    g["gate_flips"] += (_accuracy(span, ed, lo) > min_acc) != (_accuracy(span, ed, hi) > min_acc)
                    g["max_acc_spread"] = max(g["max_acc_spread"], _accuracy(span, ed, hi) - _accuracy(span, ed, lo))
                    g["max_score_spread"] = max(g["max_score_spread"], (_accuracy(span, ed, hi) - _accuracy(span, ed, lo)) * span)
            cache[key] = rec
        locs, ed, info = rec
        if ed == -1:
            return [0, 0], -1, 0
        span, ia, lo, hi = info
        ins = ia if mode[0] == "alpha" else lo if mode[0] == "min" else hi
        return locs, ed, _accuracy(span, ed, ins)

    saved = classify_port.approx_match
    classify_port.approx_match = approx_match
    try:
        out = {}
        for mname in ("alpha", "min", "max"):
            mode[0] = mname
            gaps, value, picked = classify_port.classify(sol, *idx, read, None, min_acc)
            out[mname] = (tuple((m.x, m.y, m.c, m.d, m.mam_id) for m in picked), value)
    finally:
        classify_port.approx_match = saved
    stats["instances"] += 1
    for mname in ("min", "max"):
        stats["chain_differs_" + mname] += out[mname][0] != out["alpha"][0]
        stats["value_differs_" + mname] += out[mname][1] != out["alpha"][1]
    stats["chain_differs_any"] += out["min"][0] != out["alpha"][0] or out["max"][0] != out["alpha"][0]
    big = any(20 * int(math.ceil(len(k[0]) / 64.0)) * v[2][0] >= HIRSCHBERG_BYTES for k, v in cache.items() if v[1] >= 0)
    stats["instances_with_hirschberg_hit"] += big
    stats["chain_differs_any_with_hirschberg_hit"] += big and (out["min"][0] != out["alpha"][0] or out["max"][0] != out["alpha"][0])


def _new_stats():
    g = lambda: {"hits": 0, "range": 0, "alpha_not_min": 0, "alpha_not_max": 0, "gate_flips": 0, "max_acc_spread": 0.0,
                 "max_score_spread": 0.0}
    s = {"instances": 0, "chain_differs_min": 0, "chain_differs_max": 0, "chain_differs_any": 0, "value_differs_min": 0,
         "value_differs_max": 0, "instances_with_hirschberg_hit": 0, "chain_differs_any_with_hirschberg_hit": 0,
         "stored": g(), "hirschberg": g()}
    return s


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict):
            _merge(a[k], v)
        elif k.startswith("max_"):
            a[k] = max(a[k], v)
        else:
            a[k] += v


def _worker(rng):
    stats = _new_stats()
    for i in rng:
        _solve(_G["insts"][i], _G["idx"], _G["min_acc"], stats)
    return stats


def measure(idx, insts, min_acc, workers=1):
    """Exposure figures over `insts` = [(read, solution, chr)]."""
    _G.update(idx=idx, insts=insts, min_acc=min_acc)
    cport.lib()
    if workers <= 1:
        tot = _worker(range(len(insts)))
    else:
        chunks = [range(j, len(insts), workers) for j in range(workers)]
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_worker, chunks)
        tot = _new_stats()
        for p in parts:
            _merge(tot, p)
    for g in ("stored", "hirschberg"):
        h = max(tot[g]["hits"], 1)
        tot[g]["share_with_range"] = tot[g]["range"] / h
        tot[g]["share_gate_could_flip"] = tot[g]["gate_flips"] / h
    n = max(tot["instances"], 1)
    tot["share_instances_chain_could_differ"] = tot["chain_differs_any"] / n
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6000, help="instances per preset")
    ap.add_argument("--world", default="c3")
    ap.add_argument("--presets", default="ont_cdna,ont_drna,isoseq")
    ap.add_argument("--workers", type=int, default=max(1, min(8, len(os.sched_getaffinity(0)))))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_parity_exposure.json"))
    args = ap.parse_args()
    from ultra_amd import synth      # the generator only (libultra_synth.so); no GPU, no libultra_hip.so
    t0 = time.time()
    w = synth.make_world(seed=20241008, **synth.WORLDS[args.world])
    idx = w.index()
    flat = synth.FlatWorld(w)
    res = {"what": "exposure of the MAM stage to edlib's unpinned traceback choice (oracle/parity_exposure.py)",
           "world": args.world, "transcripts": len(w.transcripts), "world_s": time.time() - t0, "presets": {}}
    for preset in args.presets.split(","):
        rs = synth.make_reads_native(flat, args.reads, seed=1000, preset=preset, threads=args.workers)   # batch 0 of bench.py
        insts = [rs.instance(i) for i in range(len(rs))]
        t0 = time.time()
        r = measure(idx, insts, synth.PRESETS[preset]["min_acc"], args.workers)
        r["seconds"] = time.time() - t0
        r["min_acc"] = synth.PRESETS[preset]["min_acc"]
        res["presets"][preset] = r
        print(preset, json.dumps(r), flush=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
