#!/bin/bash
# round 3, GPU pass s: parameter streams as vector segments for workgroups up to 512 threads, planes for 1 024 -- suite + A/B against planes everywhere + probes
set -u
O=$PWD/gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
OL=$PWD/gpurun_variants/oldlayout/libpbdx.so
for rep in 1 2 3; do
  run "c2 vector segments (rep $rep)" "" --workload c2 --steps 100 --warmup 30
  run "c2 planes (rep $rep)" "$OL" --workload c2 --steps 100 --warmup 30
done
for m in 2 4 6; do
  for rep in 1 2; do
    run "c3 m$m vector segments (rep $rep)" "" --workload c3 --solid-method $m --steps 30 --warmup 5
    run "c3 m$m planes (rep $rep)" "$OL" --workload c3 --solid-method $m --steps 30 --warmup 5
  done
done
run "c4 vector segments" "" --workload c4 --steps 50 --warmup 10
run "c4 planes" "$OL" --workload c4 --steps 50 --warmup 10
for sz in 100 300; do
  run "cloth $sz vector segments" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz planes" "$OL" --workload c2 --size $sz --steps 100 --warmup 20
done
run "c3 m2 16 bars vector segments" "" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c3 m2 16 bars planes" "$OL" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
PBDX_LIB=$PWD/gpurun_variants/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --bar 2 > $O/probe_bar_m2.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; head -9 $O/probe_bar_m2.log
