#!/usr/bin/env python3
"""Per-kernel SQ counters of scripts/profile_round.sh's `sq` pass -> profiles/<tag>_sq_counters.json.

    python scripts/sq_summary.py <round tag> <sq counter_collection.csv>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("myers_", "k_mam_build", "chain_")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    tag, path = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "rocprofv3 --pmc pass of `bench.py --steps 2 --warmup 1` (the default batch; kernels serialised by "
                   "the profiler). Per launch averages; launches_per_run = launches / launches of k_mam_build (one per run), "
                   "valu_insts_per_run = their product. SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over wavefronts; "
                   "GRBM_GUI_ACTIVE is summed over the 8 XCDs. valu_cycles_per_inst_per_simd = (GRBM_GUI_ACTIVE / 8) / "
                   "(SQ_INSTS_VALU / 1024 SIMDs): 2.0 would be the issue limit of plain 32-bit VALU ops "
                   "(profiles/r01_valu_issue_rate.txt).", "kernels": {}}
    runs = max([len(v.get("SQ_WAVES", [])) for k, v in acc.items() if k.startswith("k_mam_build")] or [1])
    for k in sorted(acc):
        if not k.startswith(KEEP):
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        per_run = len(acc[k].get("SQ_INSTS_VALU", [])) / float(max(runs, 1))
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        valu = c.get("SQ_INSTS_VALU", 0.0)
        out["kernels"][k] = {
            "waves": c.get("SQ_WAVES"), "launches_per_run": round(per_run, 2), "valu_insts": valu, "valu_insts_per_run": valu * per_run, "salu_insts": c.get("SQ_INSTS_SALU"), "lds_insts": c.get("SQ_INSTS_LDS"),
            "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
            "gui_active_per_xcd": gui, "valu_cycles_per_inst_per_simd": round(gui / (valu / 1024.0), 2) if valu else None}
    with open(os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    fam = [v for k, v in out["kernels"].items() if k.startswith("myers_mega<0,")]
    print("forward VALU insts per run: %.4g   all kept kernels: %.4g" % (sum(v["valu_insts_per_run"] for v in fam),
                                                                        sum(v["valu_insts_per_run"] for v in out["kernels"].values())))
    for pre in ("myers_mega<0,", "myers_mega<1,", "myers_mega<2,", "myers_band", "k_mam_build", "chain_"):
        print("  %-16s %.4g" % (pre, sum(v["valu_insts_per_run"] for k, v in out["kernels"].items() if k.startswith(pre))))


if __name__ == "__main__":
    main()
