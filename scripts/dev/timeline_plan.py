#!/usr/bin/env python3
"""Kernel timeline of the last planner call (k_presence .. the step's first forward kernel) from a rocprofv3 --kernel-trace CSV,
with the idle time in front of every kernel."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
fw = [i for i, r in enumerate(tr) if 'myers_mega<0' in r['Kernel_Name']]
starts = [i for i in fw if i == 0 or 'myers_mega<0' not in tr[i - 1]['Kernel_Name']]
i1 = starts[int(sys.argv[2]) if len(sys.argv) > 2 else -1]   # which step's planner, counted from the end
i0 = max(i for i, r in enumerate(tr[:i1]) if 'k_presence' in r['Kernel_Name'])
t0 = int(tr[i0]['Start_Timestamp'])
end = t0
for r in tr[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    print("%8.3f +%7.3f ms  (idle %6.3f)  %s" % ((s - t0) / 1e6, (e - s) / 1e6, max(0, s - end) / 1e6, n))
    end = max(end, e)
