"""Summarises a rocprofv3 output tree written by profiles/pmc_passes.sh into
one JSON: per-kernel average duration and per-launch counter means."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("rt::", "")


def main(root, out):
    res = {"kernels": {}, "counters": defaultdict(dict)}
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            res["kernels"][short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                                 "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                                                 "pct": float(row["Percentage"])}
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res["counters"][k][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
