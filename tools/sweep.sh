#!/bin/bash
# Sweeps the phased kernel's scheduling thresholds on the bench workload (run on the GPU box).
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  args=""
  for kv in ${cfg//,/ }; do args="$args --opt $kv"; done
  out=$(timeout -k 10 120 python bench.py --traversal 2 --no-cpu --steps 10 $args 2>/dev/null)
  echo "$cfg $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')"
done
