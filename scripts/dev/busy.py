#!/usr/bin/env python3
"""GPU busy fraction (union of kernel intervals) over the last `secs` seconds of a rocprofv3 --kernel-trace CSV."""
import csv, sys
tr = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
tr.sort()
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
end = max(e for _, e, _ in tr)
lo = end - int(secs * 1e9)
iv = [(max(s, lo), e) for s, e, _ in tr if e > lo]
iv.sort()
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.3f s: some kernel running %.1f %% of the time" % (secs, 100.0 * busy / (end - lo)))
tot = {}
for s, e, n in tr:
    if e > lo:
        k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
        tot[k] = tot.get(k, 0) + (e - max(s, lo))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-42s %8.1f ms" % (k, v / 1e6))
