#!/bin/bash
# Evidence run on the GPU box: rocprofv3 kernel stats of the bench command, counter passes for configurations 2 and
# 4, the gather microbenchmark.  Usage: tools/gpu_evidence.sh <dir under gpurun_out>.  Every GPU step has its own
# time limit and the chain stops at the first failure.
set -u
D="$1"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$D"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 ./tools/micro/tcp_gather > "$OUT/tcp_gather.txt" 2>&1 &&
timeout -k 10 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" &&
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu --no-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err") &&
tools/prof_pmc.sh "$D/pmc" &&
python tools/pmc_summary.py "$OUT/pmc" k_trace_phased > "$OUT/pmc_summary.txt" &&
tools/prof_pmc.sh "$D/pmc_c4" tools/c4_prof.py 1 &&
python tools/pmc_summary.py "$OUT/pmc_c4" k_trace_phased > "$OUT/pmc_summary_c4.txt" &&
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \; &&
(find "$OUT/stats" -name "*kernel_trace.csv" -exec head -40 {} \; > "$OUT/kernel_trace_head.csv") &&
rm -rf "$OUT/stats" && find "$OUT" -name "*.db" -delete && echo "evidence done"
