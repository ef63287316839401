#!/bin/bash
# wave-scoped colour synchronisation, first run: parity subset, then A/B against the workgroup barrier (gpurun_variants/barrier: -DPBDX_WAVE_SYNC=0)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_full_size_c2_hundred_steps_and_f64_envelope \
  --deselect tests/test_gpu_parity.py::test_bounds_checked_build_finds_no_out_of_range_access > gpurun_out/r06m_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r06m_pytest.log
V=$PWD/gpurun_variants/barrier/libpbdx.so
bash scripts/ab.sh --reps 2 --arm "wave sync" --arm "barrier, whole-step order:PBDX_LIB=$V PBDX_PLAN_BANK_WINDOW=0" --arm "barrier, wave windows:PBDX_LIB=$V" -- "" "--workload c4" "--workload c3" 2>&1 | tee gpurun_out/r06m_ab.log
