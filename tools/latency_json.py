#!/usr/bin/env python3
"""Writes latency_<tag>.json for bench.py's Little's-law ceiling from the counter passes of tools/prof_latency.sh:
the mean latency of a vector-memory instruction of one kernel (rocprofv3's derived VmemLatency, or
SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM from the same runs when the derived pass is missing).

    tools/latency_json.py <dir with the passes> <workload key> <kernel> <waves per SIMD> <out.json> [clock GHz]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def means(root, kernel):
    totals = defaultdict(list)
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel in row["Kernel_Name"]:
                    per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        for d in per.values():
            for k, v in d.items():
                totals[k].append(v)
    return {k: sum(v) / len(v) for k, v in totals.items()}


def main():
    root, workload, kernel, waves, out = sys.argv[1:6]
    clock = float(sys.argv[6]) if len(sys.argv) > 6 else 2.4
    m = means(root, kernel)
    if "VmemLatency" in m:
        latency, how = m["VmemLatency"], "rocprofv3 derived counter VmemLatency"
    else:
        latency, how = m["SQ_INST_LEVEL_VMEM"] / m["SQ_INSTS_VMEM"], "SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM"
    rec = {"workload": workload, "kernel": kernel, "vmem_latency_cycles": round(latency, 1), "how": how,
           "waves_per_simd": int(waves), "clock_ghz": clock,
           "wait_share": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m else None,
           "counters": {k: m[k] for k in sorted(m)}}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
