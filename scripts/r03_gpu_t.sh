#!/bin/bash
# round 3, GPU pass t: LDS gather issued ahead of the record fetch (workgroups up to 512 threads) -- suite + A/B + probes
set -u
O=$PWD/gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NG=$PWD/gpurun_variants/noga/libpbdx.so
for m in 2 4 6; do
  for rep in 1 2; do
    run "c3 m$m gather ahead (rep $rep)" "" --workload c3 --solid-method $m --steps 30 --warmup 5
    run "c3 m$m fetch after the barrier (rep $rep)" "$NG" --workload c3 --solid-method $m --steps 30 --warmup 5
  done
done
for sz in 100 200 300; do
  run "cloth $sz gather ahead" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz fetch after the barrier" "$NG" --workload c2 --size $sz --steps 100 --warmup 20
done
run "c3 m2 16 bars gather ahead" "" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c3 m2 16 bars fetch after the barrier" "$NG" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c2 (1024 threads: unchanged code)" "" --workload c2 --steps 100 --warmup 30
PBDX_LIB=$PWD/gpurun_variants/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --bar 2 > $O/probe_bar_m2.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; head -9 $O/probe_bar_m2.log
