#!/bin/bash
# The root rank's side of an N-GPU gather rehearsed on one GPU (bench.py --force-dist --loopback --fanin K): how the
# rate changes with the number of frames received per step and with the four-wave groups left free for RCCL's kernels.
# Usage: tools/fanin_sweep.sh
cd "$GRAFT_REPO_ROOT"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per step, kernel', d['roofline']['kernel_ms'], 'ms')"; }
for mb in 1248 1152 1024 896; do
  timeout -k 10 120 python bench.py --force-dist --loopback --fanin 8 --no-cpu --no-configs --opt max_blocks=$mb 2>/dev/null | show "fan-in 8, max_blocks $mb:" || exit 1
done
for mb in 1280 1248 1024; do
  timeout -k 10 120 python bench.py --no-cpu --no-configs --opt max_blocks=$mb 2>/dev/null | show "no gather, max_blocks $mb:" || exit 1
done
