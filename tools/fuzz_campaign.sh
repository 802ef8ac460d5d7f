#!/bin/bash
# The randomised GPU tests with many seeds (they take their seed count from RTG_FUZZ_SEEDS), one after the other;
# stops at the first failing step.
#   tools/fuzz_campaign.sh OUT_DIR SCENES PHOTON_SCENES TREES
set -u
out=${1:-gpurun_out/fuzz}
scenes=${2:-3000}
photon=${3:-300}
trees=${4:-1500}
mkdir -p "$out"
log="$out/fuzz_campaign.txt"
: > "$log"
step() {  # label, time limit, command...
  local label=$1 limit=$2
  shift 2
  echo "$label" | tee -a "$log"
  timeout -k 10 "$limit" "$@" > "$out/step.log" 2>&1
  local rc=$?
  tail -1 "$out/step.log" | tee -a "$log"
  if [ $rc -ne 0 ]; then echo "FAILED (exit $rc)" | tee -a "$log"; tail -30 "$out/step.log"; exit 1; fi
}
export RTG_FUZZ_SEEDS=$scenes
step "RTG_FUZZ_SEEDS=$scenes pytest tests/test_gpu_parity.py -k special_values" 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k special_values
step "RTG_FUZZ_SEEDS=$scenes pytest tests/test_gpu_parity.py -k shaded" 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k shaded
step "RTG_FUZZ_SEEDS=$scenes pytest tests/test_trispline.py -k fuzz" 900 python -u -m pytest tests/test_trispline.py -m gpu -q -x -k fuzz
export RTG_FUZZ_SEEDS=$photon
step "RTG_FUZZ_SEEDS=$photon pytest tests/test_gpu_photon.py -k random" 900 python -u -m pytest tests/test_gpu_photon.py -m gpu -q -x -k random
step "tools/kd_fuzz.py $trees 11" 900 python -u tools/kd_fuzz.py "$trees" 11
