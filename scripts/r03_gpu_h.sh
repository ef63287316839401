#!/bin/bash
# round 3, GPU pass h: boundary-only write-back between the passes of the persistent launch -- suite + A/B
set -u
O=$PWD/gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
run() {
  local label="$1"; local lib="$2"; shift; shift
  echo "== $label" >> $O/ab.log
  PBDX_LIB=$lib timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-roofline --no-traffic --steps 100 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'])" >> $O/ab.log 2>&1
}
FW=$PWD/gpurun_variants/fullwb/libpbdx.so
for rep in 1 2 3; do
  run "c2 boundary-only write-back (rep $rep)" "" --workload c2
  run "c2 full write-back (rep $rep)" "$FW" --workload c2
done
run "c4 block boundary-only" "" --workload c4
run "c4 block full" "$FW" --workload c4
run "c3 m2 boundary-only" "" --workload c3 --solid-method 2
run "c3 m2 full" "$FW" --workload c3 --solid-method 2
run "c3 m6 boundary-only" "" --workload c3 --solid-method 6
run "c3 m6 full" "$FW" --workload c3 --solid-method 6
run "cloth 200 boundary-only" "" --workload c2 --size 200
run "cloth 200 full" "$FW" --workload c2 --size 200
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/rc.txt; cat $O/ab.log
