#!/bin/bash
# round 5, GPU call F: full GPU suite on the packed-record sweep (12-byte LDS stores on by default), the range-checked build on the BASELINE kernels, A/B of
# the packed records, the plug-in's speculative step
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05f; mkdir -p $O; export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -m gpu -q -x -s --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_gpu.log )
B=$PWD/positionbaseddynamics_amd/_lib/libpbdx_bounds.so
for cfg in "--sim 4 --bend 3" "--sim 1 --bend 2"; do
  pass=0; fault=0
  for i in 1 2 3 4; do
    PBDX_LIB=$B timeout 120 python scripts/dev/fault_repro.py --reps 4 --persistent 2 $cfg > $O/hunt_$i.log 2>&1
    if grep -q "Memory access fault" $O/hunt_$i.log; then fault=$((fault+1)); elif grep -q "completed" $O/hunt_$i.log; then pass=$((pass+1)); fi
  done
  echo "hunt[bounds, $cfg, forced persistent]: $pass completed, $fault memory faults of 4 processes x 4 solvers" | tee -a $O/hunt.log
done
bash scripts/ab.sh --reps 2 --arm "packed" --arm "unpacked plain:PBDX_NO_PACK=1" --arm "packed, ids off:PBDX_NO_LDS_IDS=1" -- "" "--workload c4" "--workload c3" "--workload c3 --solid-method 6" 2>&1 | tee $O/ab.log
