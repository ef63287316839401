#!/bin/bash
# round 5, GPU call H: deferred fill wait on the final sweep (A/B), step probes and tile timelines of the final kernel, single-process ensemble smoke
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05h; mkdir -p $O; export TMPDIR=/tmp
V=$PWD/gpurun_variants
bash scripts/ab.sh --reps 2 --arm "in-tree" --arm "deferred fill wait:PBDX_LIB=$V/defer/libpbdx.so" -- "" "--workload c4" 2>&1 | tee $O/ab_defer.log
[ -f $V/probe/libpbdx.so ] && PBDX_LIB=$V/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --cloth 1000 > $O/step_probes_cloth.log 2>&1
tail -n 30 $O/step_probes_cloth.log
timeout 200 python scripts/trace_tiles.py --persistent 2 > $O/trace_cloth_persistent.log 2>&1; tail -n 25 $O/trace_cloth_persistent.log
timeout 300 python bench.py --gpus 2 --single-process --devices 0,0 --workload c4 --instances 16 --steps 10 --warmup 3 > $O/bench_single_process_2x16.txt 2> $O/bench_single_process.err; tail -c 1500 $O/bench_single_process_2x16.txt; tail -n 3 $O/bench_single_process.err
