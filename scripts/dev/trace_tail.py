#!/usr/bin/env python3
"""Development aid: every kernel of a rocprofv3 --kernel-trace CSV after the N-th last k_sum_i32 (end of a MAM run), with idle gaps."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(tr) if 'k_sum_i32' in r['Kernel_Name']]
i0 = marks[-1] + 1 if marks else 0
t0 = int(tr[i0]['Start_Timestamp'])
end = t0
for r in tr[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48]
    if (e - s) / 1e6 >= 0.02 or (s - end) / 1e6 >= 0.05:
        print("%9.3f +%7.3f ms  (idle %6.3f)  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, max(0, s - end) / 1e6, r.get('Queue_Id', '?'), n))
    end = max(end, e)
