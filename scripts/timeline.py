#!/usr/bin/env python3
"""Prints the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'myers_mega<0' in r['Kernel_Name']]
# first forward kernel of the last step
last = idx[-1]
while last - 1 in idx or (last > 0 and 'myers_mega<0' in tr[last - 1]['Kernel_Name']):
    last -= 1
starts = [i for i in idx if i == 0 or 'myers_mega<0' not in tr[i - 1]['Kernel_Name']]
i0 = starts[-1]
t0 = int(tr[i0]['Start_Timestamp'])
for r in tr[i0:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d < (float(sys.argv[2]) if len(sys.argv) > 2 else 0.05):
        continue
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    print("%8.3f +%7.3f ms  %s" % ((int(r['Start_Timestamp']) - t0) / 1e6, d, n))
