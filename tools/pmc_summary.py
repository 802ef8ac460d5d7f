#!/usr/bin/env python3
"""Summarises the per-dispatch counter CSVs written by tools/prof_pmc.sh for one kernel name substring."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, kernel = sys.argv[1], sys.argv[2]
    totals = defaultdict(list)
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        for d in per.values():
            for k, v in d.items():
                totals[k].append(v)
    print(f"kernel filter: {kernel}")
    for k in sorted(totals):
        v = totals[k]
        print(f"{k:32s} dispatches={len(v):3d} mean={sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main()
