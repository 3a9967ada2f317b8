"""Development aid: per-step kernel times from a rocprofv3 kernel stats CSV.  python scripts/dev/kstats.py <kt_kernel_stats.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    t = int(r["TotalDurationNs"]) / 1e6 / steps
    tot += t
    if t > 0.02:
        print("%-72s calls/step %6.1f  ms/step %7.3f" % (n, int(r["Calls"]) / steps, t))
print("sum of kernel time per step %.2f ms" % tot)
