"""Summarise rocprofv3 --pmc CSVs of tensor builds: counter sums per kernel family (eri_team_kernel by team size, the other ERI
kernels, the slab transforms) and in total.   usage: python tools/pmc_eri_summary.py OUT.json DIR"""
import collections, csv, glob, json, os, re, sys
out, src = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"eri_team_kernel<(\d+), (\d+), (\d+)>", name)
        fam = ("eri_team_kernel/team%s" % m.group(3)) if m else name.split("(")[0].split("<")[0].replace("void ", "").replace("tfk::", "")
        if not ("eri_" in fam or "xform" in fam):
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        agg["ALL ERI+xform"][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add((f, r["Dispatch_Id"]))
res = {k: dict(v, dispatches=len(disp[k])) for k, v in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items()):
    print(k, {c: ("%.4g" % x) for c, x in v.items()})
