"""Summarise rocprofv3 --pmc counter_collection CSVs into the per-kernel JSON kept under profiles/ (mean / max per dispatch,
FETCH_SIZE and WRITE_SIZE in KB as the counters report them).
usage: python tools/pmc_summary.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"dispatches": len(v), "mean_KB": sum(v) / len(v), "max_KB": max(v)} for c, v in cs.items()} for k, cs in agg.items()}
    json.dump(res, open(out, "w"), indent=1)
    for k, cs in res.items():
        print(k, {c: round(v["mean_KB"], 1) for c, v in cs.items()})


if __name__ == "__main__":
    main()
