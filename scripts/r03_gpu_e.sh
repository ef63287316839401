#!/bin/bash
# round 3, GPU pass e: quad-lane FEM projections -- bit-identity (whole suite) and A/B against the one-lane-per-constraint build
set -u
O=$PWD/gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {  # label, lib ("" = product), args...
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 120 python bench.py "$@" --no-cpu-baseline --no-extras --no-traffic --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NQ=$PWD/gpurun_variants/noquad/libpbdx.so
for m in 2 3; do
  it=10; [ $m = 3 ] && it=1
  for tile in 0 64 91 182; do
    run "c3 method $m tile $tile QUAD" "" --workload c3 --solid-method $m --iters $it --tile $tile
    run "c3 method $m tile $tile scalar" "$NQ" --workload c3 --solid-method $m --iters $it --tile $tile
  done
done
run "c3 method 4 (strain) default, dedicated kernel" "" --workload c3 --solid-method 4
run "c3 method 6 default" "" --workload c3 --solid-method 6
run "c3 32 bars method 2 QUAD" "" --workload c3 --solid-method 2 --bars --instances 32
run "c3 32 bars method 2 scalar" "$NQ" --workload c3 --solid-method 2 --bars --instances 32
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log
