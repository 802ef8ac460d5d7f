#!/bin/bash
# Vector-memory latency of the traversal kernel: instructions in flight per cycle against instructions issued (SQ),
# and the L1's own latency sums (TCP), one rocprofv3 --pmc pass per run.
#   tools/prof_latency.sh <out_dir_under_gpurun_out> [script relative to the repo root] [script args...]
# (run on the GPU box from the repo root; default script: the bench frame)
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
if [ $# -gt 0 ]; then SCRIPT="$1"; shift; ARGS=("$@"); else SCRIPT="bench.py"; ARGS=(--steps 3 --warmup 1 --windows 1 --no-cpu --no-configs); fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run sq SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM &&
run tcp TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum &&
run derived VmemLatency || echo "derived pass not available"
echo "latency passes done in $OUT"
