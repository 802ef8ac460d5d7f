#!/bin/bash
# Everything the round's numbers come from, in one run on the GPU box: tools/gpu_evidence.sh (bench, rocprofv3 stats,
# counter passes for configurations 2 and 4, gather microbenchmark), counters of configuration 3, the vector-memory
# latency and L1-path passes, the derived ceiling / traffic / latency records, the photon timings, the default job,
# configurations 3 and 5 as programs, the distributed rehearsals and the GPU test suite.
# Usage: tools/round_evidence.sh <dir under gpurun_out> [counters|programs] (default: both, which needs about a quarter
# of an hour of GPU time).  Every step has its own time limit; the chain stops at the first failure.
set -u
D="$1"
PART="${2:-both}"
cd "$GRAFT_REPO_ROOT"
O="gpurun_out/$D"
mkdir -p "$O"
counters() {
tools/gpu_evidence.sh "$D" &&
tools/prof_pmc.sh "$D/pmc_c3" tools/c3_prof.py 2 &&
python tools/pmc_summary.py "$O/pmc_c3" k_trace_phased > "$O/pmc_summary_c3.txt" &&
tools/prof_latency.sh "$D/latency" &&
tools/prof_latency.sh "$D/latency_c4" tools/c4_prof.py 1 &&
tools/prof_latency.sh "$D/latency_c3" tools/c3_prof.py 2 &&
tools/prof_l1.sh "$D/l1" &&
python tools/pmc_summary.py "$O/l1" k_trace_phased > "$O/l1_path_counters.txt" &&
tools/prof_l2.sh "$D/l2_c4" &&
python tools/pmc_summary.py "$O/l2_c4" k_trace_phased > "$O/l2_counters_c4.txt" &&
python tools/roofline_ceiling.py "$O" config2-1920x1080 k_trace_phased 2 > /dev/null &&
python tools/roofline_ceiling.py "$O" config4-64Mrays-10Mtris k_trace_phased 2 "$O/pmc_summary_c4.txt" "$O/tcp_gather.txt" 33554432 1 > /dev/null &&
python tools/roofline_ceiling.py "$O" config3-3840x2160-patches k_trace_phased 2 "$O/pmc_summary_c3.txt" > /dev/null &&
python tools/latency_json.py "$O/latency" config2-1920x1080 k_trace_phased 5 "$O/latency.json" > /dev/null &&
python tools/latency_json.py "$O/latency_c4" config4-64Mrays-10Mtris k_trace_phased 4 "$O/latency_c4.json" > /dev/null &&
python tools/latency_json.py "$O/latency_c3" config3-3840x2160-patches k_trace_phased 3 "$O/latency_config3_3840x2160_patches.json" > /dev/null &&
find "$O" -name "*.db" -delete
}
programs() {
PHOTON_PROF_ORACLE=1 timeout -k 10 300 python tools/photon_prof.py > "$O/photon_prof.txt" 2>&1 &&
timeout -k 10 300 python tools/default_job.py 5 10000000 200000 > "$O/default_job.txt" 2>&1 &&
timeout -k 10 300 python tools/default_job.py 5 10000000 0 host > "$O/default_job_host.txt" 2>&1 &&
timeout -k 10 300 python tools/config3.py > "$O/config3.log" 2>&1 &&
timeout -k 10 300 python tools/config5.py > "$O/config5_1gpu.log" 2>&1 &&
timeout -k 10 300 python tools/c4_order.py > "$O/c4_order.txt" 2>&1 &&
timeout -k 10 200 python tools/stats.py > "$O/phase_stats.txt" 2>&1 &&
timeout -k 10 200 python bench.py --force-dist --no-cpu --no-configs > "$O/bench_forcedist.json" 2> "$O/bench_forcedist.err" &&
timeout -k 10 200 python bench.py --force-dist --loopback --no-cpu --no-configs > "$O/bench_forcedist_rccl.json" 2> "$O/bench_forcedist_rccl.err" &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu --no-configs > "$O/bench_torchrun1.json" 2> "$O/bench_torchrun1.err" &&
timeout -k 10 400 python bench.py > "$O/bench_final.json" 2> "$O/bench_final.err" &&
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > "$O/gpu_tests.txt" 2>&1
rc=$?
tail -n 2 "$O/gpu_tests.txt" 2>/dev/null
return $rc
}
rc=0
if [ "$PART" != programs ]; then counters; rc=$?; fi
if [ $rc -eq 0 ] && [ "$PART" != counters ]; then programs; rc=$?; fi
echo "round evidence ($PART) rc=$rc"
exit $rc
