#!/bin/bash
# round 3, GPU pass af: particle-id lists of the pass boundaries kept / staged in LDS (IdsInLds; PBDX_NO_LDS_IDS=1 switches it off) -- tests + A/B
set -u
O=$PWD/gpurun_out/r03af; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_plugin.py tests/test_contacts.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local off="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  if [ -n "$off" ]; then export PBDX_NO_LDS_IDS=1; else unset PBDX_NO_LDS_IDS; fi
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
for rep in 1 2 3; do
  run "c2 id lists in LDS (rep $rep)" "" --workload c2 --steps 100 --warmup 30
  run "c2 id lists from memory (rep $rep)" "1" --workload c2 --steps 100 --warmup 30
done
for m in 2 6; do
  run "c3 m$m id lists in LDS" "" --workload c3 --solid-method $m --steps 30 --warmup 5
  run "c3 m$m id lists from memory" "1" --workload c3 --solid-method $m --steps 30 --warmup 5
done
run "cloth 300 id lists in LDS" "" --workload c2 --size 300 --steps 100 --warmup 20
run "cloth 300 id lists from memory" "1" --workload c2 --size 300 --steps 100 --warmup 20
unset PBDX_NO_LDS_IDS
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -5; cat $O/rc.txt; cat $O/ab.log
