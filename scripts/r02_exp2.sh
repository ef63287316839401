#!/bin/bash
set -u
O=$PWD/gpurun_out/r02d; mkdir -p $O
for m in 2 6; do
  timeout 200 python scripts/trace_tiles.py --bar $m --persistent 2 > $O/trace_bar_m${m}_persistent.log 2>&1
  timeout 200 python scripts/trace_tiles.py --bar $m --persistent 0 > $O/trace_bar_m${m}_segments.log 2>&1
done
KERNEL=persistent_kernel OUT=$O/pmc timeout 600 bash scripts/pmc_sq.sh --workload c3 --solid-method 2 --persistent 2 --fuse 1 > $O/pmc_bar_m2.log 2>&1
KERNEL=persistent_kernel OUT=$O/pmc timeout 600 bash scripts/pmc_sq.sh --workload c3 --solid-method 6 --persistent 2 --fuse 1 > $O/pmc_bar_m6.log 2>&1
head -30 $O/trace_bar_m2_persistent.log; cat $O/pmc_bar_m2.log
