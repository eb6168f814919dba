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
    res = {}
    for k, cs in agg.items():
        res[k] = {}
        for c, v in cs.items():
            e = {"dispatches": len(v), "mean": sum(v) / len(v), "max": max(v), "sum": sum(v)}
            if c.endswith("_SIZE"):                      # FETCH_SIZE / WRITE_SIZE count KB
                e["mean_KB"], e["max_KB"] = e["mean"], e["max"]
            res[k][c] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, cs in res.items():
        print(k, {c: round(v["mean"], 1) for c, v in cs.items()})


if __name__ == "__main__":
    main()
