#!/bin/bash
# bench.py --force-dist --loopback with 2, 3 and 4 hit buffers per context, fan-in 1 and 4.
cd "$GRAFT_REPO_ROOT"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per step')"; }
for k in 1 4; do for nb in 2 3 4; do
  timeout -k 10 120 python bench.py --force-dist --loopback --fanin $k --hit-buffers $nb --no-cpu --no-configs 2>/dev/null | show "fan-in $k, $nb hit buffers:" || exit 1
done; done
