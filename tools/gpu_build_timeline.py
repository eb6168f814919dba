"""Summarise a rocprofv3 --kernel-trace CSV of tools/gpu_eri_hosttime.py: every kernel of the LAST tensor build (name, start, duration)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last build starts with the last memset / first eri kernel after a gap > 2 ms
starts = [int(r['Start_Timestamp']) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - int(rows[i - 1]['End_Timestamp']) > 1.5e6: cut = i
rows = rows[cut:]
t0 = int(rows[0]['Start_Timestamp'])
busy = 0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.08:
        print('%-60s grid %7s x %-5s start %7.2f ms dur %6.2f ms' % (r['Kernel_Name'].split('(')[0][-60:], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], (int(r['Start_Timestamp']) - t0) / 1e6, d))
print('kernels in the build: %d, span %.2f ms' % (len(rows), (int(rows[-1]['End_Timestamp']) - t0) / 1e6))
