"""CPU: the roofline of the committed bench record can be recomputed from the files under profiles/ alone -- the
microbenchmark's lane-load rates and the counter means give the ceiling, the counter traffic gives the HBM figure,
and bench.py's record follows from them with the launch duration and the visits per ray it states."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench  # noqa: E402
import roofline_ceiling  # noqa: E402

CASES = [("config2-1920x1080", 1920 * 1080, 0.0, 2097152, 72.0, lambda rec: rec["roofline"]),
         ("config4-64Mrays-10Mtris", 1 << 26, 48.0, 33554432, 72.0, lambda rec: rec["configs"]["c4"]["roofline"]),
         ("config3-3840x2160-patches", 3840 * 2160, 0.0, 2097152, 312.0, lambda rec: rec["configs"]["c3"]["roofline"])]


def _newest():
    return sorted(d for d in os.listdir(os.path.join(ROOT, "profiles"))
                  if os.path.isfile(os.path.join(ROOT, "profiles", d, "bench.json")))[-1]


def _record():
    with open(os.path.join(ROOT, "profiles", _newest(), "bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("workload,n_rays,ray_bytes,beyond,leaf_bytes,pick", CASES)
def test_ceiling_follows_from_the_microbenchmark_and_the_counters(workload, n_rays, ray_bytes, beyond, leaf_bytes, pick):
    c = bench.committed("ceiling.json", workload, "k_trace_phased")
    assert c is not None
    r1, r2, r3 = roofline_ceiling.gather_rates(os.path.join(ROOT, c["gather_lane_loads_per_s"]["source"]), beyond,
                                               c["gather_lane_loads_per_s"].get("gather_mode", 0))
    assert c["gather_lane_loads_per_s"]["beyond_l2_table_records"] == beyond
    assert (r1, r2, r3) == pytest.approx((c["gather_lane_loads_per_s"]["l1"], c["gather_lane_loads_per_s"]["l2"],
                                          c["gather_lane_loads_per_s"]["beyond_l2"]), rel=1e-9)
    assert r1 > r2 > r3 > 0
    k = roofline_ceiling.counters(os.path.join(ROOT, c["fractions"]["source"]))
    to_l2 = k["TCP_TCC_READ_REQ_sum"] / k["TCP_TOTAL_CACHE_ACCESSES_sum"]
    past_l2 = to_l2 * k["TCC_MISS_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
    f = c["fractions"]
    assert (f["l1"], f["l2"], f["beyond_l2"]) == pytest.approx((1 - to_l2, to_l2 - past_l2, past_l2), abs=2e-5)
    lane_loads = 1.0 / (f["l1"] / r1 + f["l2"] / r2 + f["beyond_l2"] / r3)
    assert c["ceiling_visits_per_s"] == pytest.approx(lane_loads / c["loads_per_visit"], rel=1e-4)


@pytest.mark.parametrize("workload,n_rays,ray_bytes,beyond,leaf_bytes,pick", CASES)
def test_committed_bench_record_follows_from_the_committed_evidence(workload, n_rays, ray_bytes, beyond, leaf_bytes, pick):
    roof = pick(_record())
    again = bench.roofline_record("k_trace_phased", workload, roof["kernel_ms"], n_rays, roof["nodes_per_ray"],
                                  roof["tris_per_ray"], ray_bytes, True, leaf_bytes=leaf_bytes)
    for part in ("gather", "hbm", "little"):
        assert 0 < roof[part]["frac"] <= 1
        assert again[part]["frac"] == pytest.approx(roof[part]["frac"], rel=2e-3)
        assert again[part]["peak"] == pytest.approx(roof[part]["peak"], rel=1e-6)
    assert again["algorithmic"]["bytes_per_ray"] == pytest.approx(roof["algorithmic"]["bytes_per_ray"], rel=2e-3)
    # the latency ceiling: resident rays over the measured latency of a vector-memory instruction
    lat = bench.committed("latency.json", workload, "k_trace_phased")
    assert roof["little"]["peak"] == pytest.approx(
        1024 * lat["waves_per_simd"] * 64 / (lat["vmem_latency_cycles"] / (lat["clock_ghz"] * 1e9)) / 1e9, rel=1e-3)
    assert lat["vmem_latency_cycles"] == pytest.approx(
        lat["counters"]["SQ_INST_LEVEL_VMEM"] / lat["counters"]["SQ_INSTS_VMEM"] if "VmemLatency" not in lat["counters"]
        else lat["counters"]["VmemLatency"], rel=1e-3)
    assert roof["frac"] == max(roof["gather"]["frac"], roof["hbm"]["frac"]) and roof["frac"] <= 1
    assert again["bound"] == roof["bound"]
    # traffic: the counter bytes of the committed pass, corrected as the file says
    t = bench.committed("traffic.json", workload, "k_trace_phased")
    assert t["hbm_bytes_per_launch"] == pytest.approx(
        (2 * t["fetch_size_kb_per_launch"] + t["write_size_kb_per_launch"]) * 1024, rel=1e-9)
    assert roof["traffic"] == pytest.approx(t["hbm_bytes_per_launch"] / 1e9, rel=1e-3)


def test_launch_duration_agrees_with_the_rocprofv3_summary_of_the_same_command():
    """kernel_stats.csv is `rocprofv3 --kernel-trace --stats` over `python3 bench.py --no-cpu --no-configs`;
    bench_under_rocprof.json is the line that very run printed.  The duration bench.py takes from the launches' own
    events and the profiler's average over the same launches (warm-up included) agree within 2 %."""
    import csv
    d = os.path.join(ROOT, "profiles", _newest())
    with open(os.path.join(d, "bench_under_rocprof.json")) as f:
        rec = json.loads(f.read().strip().splitlines()[-1])
    rows = [r for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv")))
            if "k_trace_phased<rtg::PrimarySource, false" in r["Name"]]
    assert len(rows) == 1
    assert int(rows[0]["Calls"]) == rec["warmup"] + rec["settle_steps"] + rec["steps"] * rec["windows"]
    assert float(rows[0]["AverageNs"]) / 1e6 == pytest.approx(rec["roofline"]["kernel_ms"], rel=0.02)
    # the dominant kernel: nearly all of the GPU time of the run
    assert float(rows[0]["Percentage"]) > 95
