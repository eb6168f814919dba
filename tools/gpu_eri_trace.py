"""Summarise a rocprofv3 --kernel-trace CSV: the ERI generation launches (grid, start, duration)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'eri_' in r['Kernel_Name']]
t0 = min(int(r['Start_Timestamp']) for r in rows)
t1 = max(int(r['End_Timestamp']) for r in rows)
for r in rows:
    print(r['Kernel_Name'].split('(')[0][-28:], int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'], 'start %.2f ms dur %.2f ms' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
print('span %.2f ms' % ((t1 - t0) / 1e6))
