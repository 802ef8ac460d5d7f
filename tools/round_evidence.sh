#!/bin/bash
# Everything the round's numbers come from, in one run on the GPU box: tools/gpu_evidence.sh (bench, rocprofv3 stats,
# counter passes, gather microbenchmark), then the photon timings, the default job, configurations 3 and 5, the
# distributed rehearsals and the GPU test suite.  Usage: tools/round_evidence.sh <dir under gpurun_out>.  Every step has
# its own time limit; the chain stops at the first failure.
set -u
D="$1"
cd "$GRAFT_REPO_ROOT"
O="gpurun_out/$D"
tools/gpu_evidence.sh "$D" &&
PHOTON_PROF_ORACLE=1 timeout -k 10 300 python tools/photon_prof.py > "$O/photon_prof.txt" 2>&1 &&
timeout -k 10 300 python tools/default_job.py 5 10000000 200000 > "$O/default_job.txt" 2>&1 &&
timeout -k 10 300 python tools/config3.py > "$O/config3.log" 2>&1 &&
timeout -k 10 300 python tools/config5.py > "$O/config5_1gpu.log" 2>&1 &&
timeout -k 10 200 python bench.py --force-dist --no-cpu --no-configs > "$O/bench_forcedist.json" 2> "$O/bench_forcedist.err" &&
timeout -k 10 200 python bench.py --force-dist --loopback --no-cpu --no-configs > "$O/bench_forcedist_rccl.json" 2> "$O/bench_forcedist_rccl.err" &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu --no-configs > "$O/bench_torchrun1.json" 2> "$O/bench_torchrun1.err" &&
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > "$O/gpu_tests.txt" 2>&1
rc=$?
tail -n 2 "$O/gpu_tests.txt" 2>/dev/null
echo "round evidence rc=$rc"
exit $rc
