"""Summarise one SCF iteration from a rocprofv3 --hip-trace --kernel-trace run (directory given): kernels between two
consecutive density-change reductions, GPU busy time, HIP API calls."""
import collections, csv, glob, sys
d = sorted(glob.glob(sys.argv[1] + '/*/'))[-1]
api = list(csv.DictReader(open(glob.glob(d + '*hip_api_trace.csv')[0])))
ker = list(csv.DictReader(open(glob.glob(d + '*kernel_trace.csv')[0])))
ker.sort(key=lambda r: int(r['Start_Timestamp']))
dn = [i for i, r in enumerate(ker) if 'k_delta_norms' in r['Kernel_Name'] and 'fin' not in r['Kernel_Name']]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
i0, i1 = dn[-back - 1], dn[-back]
t0, t1 = int(ker[i0]['End_Timestamp']), int(ker[i1]['End_Timestamp'])
busy = 0
agg = collections.OrderedDict()
for r in ker[i0 + 1:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    n = r['Kernel_Name'].split('(')[0][-40:]
    c = agg.setdefault(n, [0, 0])
    c[0] += 1; c[1] += e - s
print('iteration wall %.3f ms, gpu busy %.3f ms, %d kernels' % ((t1 - t0) / 1e6, busy / 1e6, i1 - i0))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  %-40s x%-3d %.3f ms' % (n, c, t / 1e6))
c = collections.Counter(); dur = collections.Counter()
for a in api:
    s = int(a['Start_Timestamp'])
    if t0 <= s < t1:
        c[a['Function']] += 1; dur[a['Function']] += int(a['End_Timestamp']) - s
for k, v in c.most_common(6):
    print('  api %-28s x%-3d %.3f ms' % (k, v, dur[k] / 1e6))
