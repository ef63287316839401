#!/bin/bash
# is it the record fetch that the wave-major groups make slower?  no-fetch knock-outs (timing only) of both forms
export TMPDIR=/tmp
V=$PWD/gpurun_variants
bash scripts/ab.sh --reps 2 --arm "barrier (round 5 form):PBDX_LIB=$V/barrier/libpbdx.so" --arm "barrier, no fetch:PBDX_LIB=$V/bar_nofetch/libpbdx.so" --arm "wave-major + barrier:PBDX_LIB=$V/ws_barrier/libpbdx.so" \
  --arm "wave-major + barrier, no fetch:PBDX_LIB=$V/wsb_nofetch/libpbdx.so" -- "" 2>&1 | tee gpurun_out/r06o_ab.log
