#!/bin/bash
# round 3, GPU pass p: where one colour step of the bar goes (probe build)
set -u
O=$PWD/gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
PB=$PWD/gpurun_variants/probe/libpbdx.so
for m in 2 4 6; do
  PBDX_LIB=$PB timeout 300 python scripts/probe_steps.py --bar $m > $O/probe_bar_m$m.log 2>&1
done
cat $O/probe_bar_m2.log; tail -40 $O/probe_bar_m4.log $O/probe_bar_m6.log
