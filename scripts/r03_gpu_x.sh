#!/bin/bash
# round 3, GPU pass x: TIMING EXPERIMENT (wrong results by construction): what the 1 M cloth would cost if a bending record's ten parameter planes were ONE load
set -u
O=$PWD/gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
OP=$PWD/gpurun_variants/oneplane/libpbdx.so
for rep in 1 2; do
  run "c2 product (rep $rep)" "" --workload c2 --steps 60 --warmup 20
  run "c2 one plane instead of ten (rep $rep)" "$OP" --workload c2 --steps 60 --warmup 20
done
run "c4 product" "" --workload c4 --steps 30 --warmup 10
run "c4 one plane instead of ten" "$OP" --workload c4 --steps 30 --warmup 10
cat $O/ab.log
