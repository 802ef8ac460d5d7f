#!/usr/bin/env python3
"""Writes ceiling.json and traffic.json for one workload from committed evidence:

  tools/roofline_ceiling.py <profiles dir> <workload key> <kernel> <loads per visit> [pmc summary file] [gather file]
                            [records of the table that stands for "beyond L2": 2097152 (64 MB, default) or 33554432 (1 GB)]
                            [gather mode of the microbenchmark: 0 (default) or 1 (lane pairs fetch each record together,
                            as the kernel does for ray batches)]

* `<dir>/tcp_gather.txt` -- output of tools/micro/tcp_gather (mode 0: every lane gathers its own 32-byte record with
  two 16-byte loads, 16 waves per CU), one line per table size: 8 KB (served by the CU's vector L1), 2 MB (the XCD's
  L2), 64 MB (the Infinity Cache), 1 GB (HBM behind the Infinity Cache).  The rate used is chip-wide lane loads per
  second, as measured (no clock assumed).  A tree that fits the Infinity Cache (config 2: 144 MB) is priced with the
  64 MB table for what leaves L2, a tree that does not (config 4: 1.4 GB) with the 1 GB table.
* `<dir>/pmc_summary.txt` -- per-launch counter means of the kernel (tools/pmc_summary.py over tools/prof_pmc.sh):
  TCP_TOTAL_CACHE_ACCESSES / TCP_TCC_READ_REQ give the share of vector-L1 accesses that go on to L2, TCC_HIT / TCC_MISS
  the share of those that leave L2.

Model: a node visit costs `loads per visit` gathered lane loads; the CU's vector memory path serves lane loads at
rate r1 when they hit L1, r2 when L2 serves them, r3 beyond L2, and a miss occupies the path for the slower level's
time, so with the kernel's own fractions f1, f2, f3 the path sustains at most 1 / (f1/r1 + f2/r2 + f3/r3) lane loads
per second, and the kernel at most that / loads-per-visit node visits per second.  Triangle fetches and stores ride
the same path and are not charged, so the ceiling is on the generous side."""
import json
import os
import re
import sys


def gather_rates(path, beyond=2097152, mode=0):
    rates = {}
    for line in open(path):
        m = re.match(r"\s*(\d+) records, mode %d: ([\d.]+) ms" % mode, line)
        if m:
            n, ms = int(m.group(1)), float(m.group(2))
            lane_loads = 1024.0 * 256 * 2000 * 2
            rates[n] = max(rates.get(n, 0.0), lane_loads / (ms * 1e-3))      # the warm repetition
    return rates[256], rates[65536], rates[beyond]


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+dispatches=\s*\d+\s+mean=([\d.e+]+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def main():
    d, workload, kernel, loads = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    pmc = sys.argv[5] if len(sys.argv) > 5 else os.path.join(d, "pmc_summary.txt")
    gather = sys.argv[6] if len(sys.argv) > 6 else os.path.join(d, "tcp_gather.txt")
    beyond = int(sys.argv[7]) if len(sys.argv) > 7 else 2097152
    mode = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    r1, r2, r3 = gather_rates(gather, beyond, mode)
    c = counters(pmc)
    to_l2 = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    l2_miss = c["TCC_MISS_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    f1, f2, f3 = 1.0 - to_l2, to_l2 * (1.0 - l2_miss), to_l2 * l2_miss
    lane_loads_per_s = 1.0 / (f1 / r1 + f2 / r2 + f3 / r3)
    ceiling = {"workload": workload, "kernel": kernel, "loads_per_visit": loads,
               "gather_lane_loads_per_s": {"l1": r1, "l2": r2, "beyond_l2": r3, "beyond_l2_table_records": beyond, "gather_mode": mode,
                                           "source": os.path.relpath(gather)},
               "fractions": {"l1": round(f1, 5), "l2": round(f2, 5), "beyond_l2": round(f3, 5), "source": os.path.relpath(pmc)},
               "ceiling_lane_loads_per_s": lane_loads_per_s, "ceiling_visits_per_s": lane_loads_per_s / loads,
               "model": "1 / (f1/r1 + f2/r2 + f3/r3) / loads_per_visit"}
    suffix = "" if "config2" in workload else ("_c4" if "config4" in workload else "_" + re.sub(r"\W+", "_", workload))
    with open(os.path.join(d, f"ceiling{suffix}.json"), "w") as f:
        json.dump(ceiling, f, indent=1)
    kb = 1024.0
    traffic = {"workload": workload, "kernel": kernel, "source": os.path.relpath(pmc),
               "fetch_size_kb_per_launch": c["FETCH_SIZE"], "write_size_kb_per_launch": c["WRITE_SIZE"],
               "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request, doubled; WRITE_SIZE taken as is; counter unit KB (1024 B)",
               "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * kb}
    if "TCC_EA0_RDREQ_128B_sum" in c:
        # the doubling checked on this kernel's own access pattern: read requests to the fabric by size
        n128, n64, n_all = c["TCC_EA0_RDREQ_128B_sum"], c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c["TCC_EA0_RDREQ_sum"]
        traffic["read_requests"] = {"all": n_all, "of_128_bytes": n128, "of_64_bytes": n64,
                                    "bytes_by_request_size": 128.0 * n128 + 64.0 * n64 + 32.0 * max(0.0, n_all - n128 - n64),
                                    "fetch_size_doubled_bytes": 2.0 * c["FETCH_SIZE"] * kb}
    with open(os.path.join(d, f"traffic{suffix}.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(ceiling, indent=1))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
