#!/usr/bin/env python3
"""Summarises rocprofv3 output into profiles/: per-kernel time (kernel-trace --stats) and HBM traffic
(separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same bench command).

    python scripts/pmc_summary.py <round tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv>

FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md (section HBM): on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced streaming read and is uncalibrated for other widths; both the raw and
the doubled figure are kept, `hbm_bytes_per_launch` uses the doubled read side (upper bound).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    dominant = sys.argv[5] if len(sys.argv) > 5 else None
    rows = list(csv.DictReader(open(stats)))
    f, w = counters(fetch), counters(write)
    import datetime
    out = {"tag": tag, "recorded": datetime.date.today().isoformat(),
           "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, GPU_MAX_HW_QUEUES=%s) -- " % os.environ.get("ULTRA_HW_QUEUES", "13 (ultra_amd.hwq.DEFAULT; scripts/profile_round.sh exports it under both names)") + ""
                      "python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --check 0 --e2e-rounds 0 --front-rounds 0 --micro 0 (scripts/profile_round.sh)",
           "kernels": []}
    for r in rows:
        k = short(r["Name"])
        out["kernels"].append({"kernel": k, "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                               "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"]),
                               "fetch_kb_raw": f.get(k), "write_kb": w.get(k)})
    if dominant is None:
        fw = [k for k in out["kernels"] if k["kernel"].startswith("myers_mega<0,")]
        dominant = max(fw, key=lambda k: k["total_ms"])["kernel"]
    d = [k for k in out["kernels"] if k["kernel"] == dominant][0]
    out["dominant_kernel"] = dominant
    out["dominant_avg_us"] = d["avg_us"]
    out["hbm_bytes_per_launch"] = int(((d["fetch_kb_raw"] or 0) * 2 + (d["write_kb"] or 0)) * 1024)
    out["hbm_bytes_per_launch_raw_counters"] = int(((d["fetch_kb_raw"] or 0) + (d["write_kb"] or 0)) * 1024)
    # the forward pass = the myers_mega<0, bucket> launches of one step, which run concurrently (one per lane)
    fam = [k for k in out["kernels"] if k["kernel"].startswith("myers_mega<0,")]
    out["forward_family"] = {
        "kernels": [k["kernel"] for k in fam],
        "hbm_bytes_per_step": int(sum(((k["fetch_kb_raw"] or 0) * 2 + (k["write_kb"] or 0)) * 1024 for k in fam)),
        "longest_avg_us": max([k["avg_us"] for k in fam] or [0.0]),
        "sum_avg_us": sum(k["avg_us"] for k in fam)}
    with open(os.path.join(ROOT, "profiles", "%s_summary.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as fh:
        json.dump({k: out[k] for k in ("tag", "recorded", "command", "dominant_kernel", "dominant_avg_us", "hbm_bytes_per_launch",
                                      "hbm_bytes_per_launch_raw_counters", "forward_family")}, fh, indent=1)
    print(dominant, d["avg_us"], out["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
