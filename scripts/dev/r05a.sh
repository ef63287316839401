#!/bin/bash
# round 5, GPU call A: suite on the bank-aware planner + the bounds-checked build, fault hunt of the pipelined fetch descriptor, first A/B of the
# bank-aware slot order and the 12-byte LDS stores
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05a; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest_gpu.log )
V=$PWD/gpurun_variants
bash scripts/dev/fault_hunt.sh 12 $O/hunt in-tree=$PWD/positionbaseddynamics_amd/_lib/libpbdx.so pf=$V/pf/libpbdx.so pfdrain=$V/pfdrain/libpbdx.so pfbounds=$V/pfbounds/libpbdx.so 2>&1 | tee $O/fault_hunt.log
bash scripts/ab.sh --reps 2 --arm "bank-aware" --arm "id-order:PBDX_NO_BANK_ORDER=1" --arm "st96:PBDX_LIB=$V/st96/libpbdx.so" --arm "st96 id-order:PBDX_LIB=$V/st96/libpbdx.so PBDX_NO_BANK_ORDER=1" -- "" "--workload c4" 2>&1 | tee $O/ab.log
