#!/bin/bash
# where the wave-scoped synchronisation loses: nobody waits / wave-major groups under the barrier / polls with a sleep / one compare per poll
export TMPDIR=/tmp
V=$PWD/gpurun_variants
bash scripts/ab.sh --reps 2 --arm "barrier (round 5 form):PBDX_LIB=$V/barrier/libpbdx.so" --arm "wave-major + barrier:PBDX_LIB=$V/ws_barrier/libpbdx.so" --arm "wave sync, nobody waits:PBDX_LIB=$V/ws_nowait/libpbdx.so" \
  --arm "wave sync:PBDX_LIB=$V/ws_poll1/libpbdx.so" --arm "wave sync, s_sleep 1:PBDX_LIB=$V/ws_sleep1/libpbdx.so" -- "" "--workload c3" 2>&1 | tee gpurun_out/r06n_ab.log
