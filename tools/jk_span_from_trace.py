"""From a `rocprofv3 --kernel-trace` CSV: time per Fock build of jk_packed_kernel.  A build launches the kernel up to three times (one
launch per workgroup size) on concurrent streams, so the figure that corresponds to bench.py's `roofline.kernel_avg_ms` (HIP events
around the group on the caller's stream) is the SPAN first start -> last end of each group, not the sum of the durations.
usage: python tools/jk_span_from_trace.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
groups, cur = [], []
for s, e, name in rows:
    if "jk_packed_kernel" in name:
        cur.append((s, e))
    elif "jk_reduce_kernel" in name and cur:                 # the reduction follows every build
        groups.append(cur)
        cur = []
spans = [max(e for _, e in g) - min(s for s, _ in g) for g in groups]
sums = [sum(e - s for s, e in g) for g in groups]
n = len(groups)
print(f"{f}: {n} builds, launches per build {sum(len(g) for g in groups) / max(n, 1):.2f}")
print(f"span per build: mean {sum(spans) / n / 1e6:.4f} ms (min {min(spans) / 1e6:.4f}, max {max(spans) / 1e6:.4f}); "
      f"sum of launch durations per build: mean {sum(sums) / n / 1e6:.4f} ms")
