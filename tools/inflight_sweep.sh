#!/bin/bash
# Frames in flight per GPU on the shipped kernel: python bench.py with --inflight 1..4, one line each.
#   tools/inflight_sweep.sh OUT_DIR
set -u
out=${1:-gpurun_out/inflight}
mkdir -p "$out"
for k in 1 2 3 4; do
  timeout -k 10 120 python bench.py --no-cpu --no-configs --inflight $k > "$out/inflight$k.json" 2> "$out/inflight$k.err" || exit 1
  python - "$out/inflight$k.json" $k <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("inflight", sys.argv[2], r["value"], "Mrays/s", r["ms_per_step"], "ms/step", "kernel_ms", r["roofline"].get("kernel_ms"), flush=True)
PY
done
