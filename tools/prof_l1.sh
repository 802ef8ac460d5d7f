#!/bin/bash
# How busy the vector-memory path of the traversal kernel is: texture-addresser (TA), L1 (TCP) and data-return (TD)
# busy and stall cycles, and the SQ's issue counters by instruction type, one rocprofv3 --pmc pass per run.
#   tools/prof_l1.sh <out_dir_under_gpurun_out> [script relative to the repo root] [script args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
if [ $# -gt 0 ]; then SCRIPT="$1"; shift; ARGS=("$@"); else SCRIPT="bench.py"; ARGS=(--steps 3 --warmup 1 --windows 1 --no-cpu --no-configs); fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run ta1 TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE &&
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
run ta3 TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum &&
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_LFIFO_STALL_CYCLES_sum &&
run tcp3 TCP_TA_TCP_STATE_READ_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_RFIFO_STALL_CYCLES_sum &&
run td1 TD_TD_BUSY_sum TD_TC_STALL_sum &&
run td2 TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum &&
run sq1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run sq2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM
echo "l1 passes done in $OUT"
