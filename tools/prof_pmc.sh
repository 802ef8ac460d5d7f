#!/bin/bash
# Collects rocprofv3 counters for one kernel, one --pmc pass per run (the tool does not split counters over
# passes).  Usage: tools/prof_pmc.sh <out_dir_under_gpurun_out> [script relative to the repo root] [script args...]
# Default script: bench.py --steps 3 --warmup 1 --windows 1 --no-cpu --no-configs.  Run on the GPU box from the repo root.
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
if [ $# -gt 0 ]; then SCRIPT="$1"; shift; ARGS=("$@"); else SCRIPT="bench.py"; ARGS=(--steps 3 --warmup 1 --windows 1 --no-cpu --no-configs); fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD &&
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE &&
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
# read requests to the fabric by size (32 B, 128 B; the rest are 64 B): checks FETCH_SIZE's 64 B per request
run tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum || echo "request-size pass not available"
echo "pmc passes done in $OUT"
