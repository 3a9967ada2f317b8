#!/usr/bin/env python3
"""Every kernel and memory copy of the last bench step, in start order, with the idle time before each
(rocprofv3 --kernel-trace --memory-copy-trace CSVs).  usage: gaps.py <kernel_trace.csv> [<memory_copy_trace.csv>]"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'rocprim' in n:
        n = 'rocprim:' + ('onesweep' if 'onesweep' in n else 'merge' if 'merge' in n else 'block_sort' if 'block_sort' in n else 'transform' if 'transform' in n else 'other')
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n[:40]))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[:30]))
ev.sort()
fw = [i for i, e in enumerate(ev) if e[2].startswith('myers_mega<0')]
starts = [i for i in fw if i == 0 or not ev[i - 1][2].startswith('myers_mega<0')]
i0 = starts[-1]
t0 = ev[i0][0]
busy_end = t0
for s, e, n in ev[i0:]:
    gap = (s - busy_end) / 1e3
    print("%9.3f ms  +%8.1f us  idle before %7.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, gap if gap > 0 else 0.0, n))
    busy_end = max(busy_end, e)
