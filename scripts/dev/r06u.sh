#!/bin/bash
# the record ring's first two records requested during the fill (EarlyRing): 56 spilled VGPRs in the 1 024-thread kernel -- does it pay anyway?
export TMPDIR=/tmp
V=$PWD/gpurun_variants
bash scripts/ab.sh --reps 2 --arm "in-tree" --arm "ring requested during the fill:PBDX_LIB=$V/early2/libpbdx.so" -- "" "--workload c4" "--workload c3" "--size 300" 2>&1 | tee gpurun_out/r06u_early_ring.log
