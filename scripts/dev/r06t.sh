#!/bin/bash
# configs[3] block: where does a pass of the walk go (pass probes), and does a deeper record ring help where the fetch is not hidden (no-fetch bound -6.7 %)?
export TMPDIR=/tmp
V=$PWD/gpurun_variants
PBDX_LIB=$V/passprobe/libpbdx.so timeout 600 python scripts/probe_pass.py --size 200 --instances 64 2>&1 | tee gpurun_out/r06t_pass_probes_c4.log | tail -45
bash scripts/ab.sh --reps 2 --arm "in-tree" --arm "ring depth 4 for the wide records:PBDX_LIB=$V/depth4/libpbdx.so" -- "" "--workload c4" 2>&1 | tee gpurun_out/r06t_depth4.log
