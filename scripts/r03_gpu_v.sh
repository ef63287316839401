#!/bin/bash
# round 3, GPU pass v: fetch descriptor read before the projection (workgroups up to 512 threads) -- A/B (+ quick parity subset)
set -u
O=$PWD/gpurun_out/r03v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NH=$PWD/gpurun_variants/nohoist/libpbdx.so
for m in 2 4 6; do
  for rep in 1 2; do
    run "c3 m$m descriptor hoisted (rep $rep)" "" --workload c3 --solid-method $m --steps 30 --warmup 5
    run "c3 m$m descriptor after the barrier (rep $rep)" "$NH" --workload c3 --solid-method $m --steps 30 --warmup 5
  done
done
for sz in 100 200 300; do
  run "cloth $sz descriptor hoisted" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz descriptor after the barrier" "$NH" --workload c2 --size $sz --steps 100 --warmup 20
done
run "c3 m2 16 bars descriptor hoisted" "" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c3 m2 16 bars descriptor after the barrier" "$NH" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
grep -E "passed|failed" $O/pytest.log | tail -3; cat $O/rc.txt; cat $O/ab.log
