#!/bin/bash
# round 3, GPU pass q: waves without slots in a chunk skip its record fetch (small-scene kernels) -- suite + A/B + probes
set -u
O=$PWD/gpurun_out/r03q; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
NS=$PWD/gpurun_variants/noskip/libpbdx.so
for m in 2 4 6; do
  for rep in 1 2; do
    run "c3 m$m skip idle waves (rep $rep)" "" --workload c3 --solid-method $m --steps 30 --warmup 5
    run "c3 m$m all waves fetch (rep $rep)" "$NS" --workload c3 --solid-method $m --steps 30 --warmup 5
  done
done
for sz in 100 200 300; do
  run "cloth $sz skip" "" --workload c2 --size $sz --steps 100 --warmup 20
  run "cloth $sz all fetch" "$NS" --workload c2 --size $sz --steps 100 --warmup 20
done
run "c3 m2 16 bars skip" "" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c3 m2 16 bars all fetch" "$NS" --workload c3 --solid-method 2 --bars --instances 16 --steps 20 --warmup 5
run "c5 skip" "" --workload c5
PBDX_LIB=$PWD/gpurun_variants/probe/libpbdx.so timeout 300 python scripts/probe_steps.py --bar 2 > $O/probe_bar_m2.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log; head -12 $O/probe_bar_m2.log
