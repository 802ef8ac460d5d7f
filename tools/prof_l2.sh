#!/bin/bash
# What the L2, the fabric requests and the address translation of the traversal kernel look like when the rays are
# unrelated: L2 busy and stall cycles, depth of the read queue towards memory (queue depth / requests = latency in L2
# clocks), credit stalls, L1 -> L2 read latency, TLB hits and misses.  One rocprofv3 --pmc pass per run.
#   tools/prof_l2.sh <out_dir_under_gpurun_out> [script relative to the repo root] [script args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
if [ $# -gt 0 ]; then SCRIPT="$1"; shift; ARGS=("$@"); else SCRIPT="tools/c4_prof.py"; ARGS=(1); fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE &&
run tcc2 TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_LATENCY_FIFO_FULL_sum &&
run tcc3 TCC_HIT_sum TCC_MISS_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum &&
run tcp1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum &&
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
run tcp3 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_LFIFO_FULL_sum
echo "l2 passes done in $OUT"
